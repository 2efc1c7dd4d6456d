# SQ counters of the boundary / statistics kernels on config 4's windows -> gpurun_out/pmc_boundary/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_boundary
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU_TRANS_F64"
timeout -s KILL 120 rocprofv3 --pmc $A -d $R/gpurun_out/pmc_boundary/a -o a --output-format csv -- python3 $R/tools/exp_boundary.py 2000 50000 2 > $R/gpurun_out/pmc_boundary/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $B -d $R/gpurun_out/pmc_boundary/b -o b --output-format csv -- python3 $R/tools/exp_boundary.py 2000 50000 2 > $R/gpurun_out/pmc_boundary/b.log 2>&1
