# SQ_INSTS_VALU (and the FP64 classes) of the subsystem kernels bench.py times after its steps -> gpurun_out/pmc_side/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_side
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
timeout -s KILL 120 rocprofv3 --pmc $A -d $R/gpurun_out/pmc_side/a -o a --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_side/a.log 2>&1
ls $R/gpurun_out/pmc_side/a
