# bench.py once, then rocprofv3: kernel trace with stats, and the two HBM counter passes, each in
# a run of its own.  Outputs under gpurun_out/$1/ ; tools/summarize_profile.py turns them into profiles/.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout -k 10 300 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 2500 gpurun_out/$TAG/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o t --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/$TAG/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$TAG/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$TAG/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/$TAG/pmc_write.log 2>&1
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC"
timeout -s KILL 120 rocprofv3 --pmc $A -d $R/gpurun_out/$TAG/sq_a -o a --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/$TAG/sq_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $B -d $R/gpurun_out/$TAG/sq_b -o b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/$TAG/sq_b.log 2>&1
find $R/gpurun_out/$TAG -name "*.csv" | head -20
