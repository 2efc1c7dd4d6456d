# SQ counters of the bench kernels: separate passes of at most 8 SQ counters each -> gpurun_out/pmc_sq/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_sq
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC"
C="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAVES"
timeout -s KILL 90 rocprofv3 --pmc $A -d $R/gpurun_out/pmc_sq/a -o a --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/pmc_sq/a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc $B -d $R/gpurun_out/pmc_sq/b -o b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/pmc_sq/b.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_sq/c -o c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-side > $R/gpurun_out/pmc_sq/c.log 2>&1
ls $R/gpurun_out/pmc_sq/*
