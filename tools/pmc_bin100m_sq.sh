# SQ counters of the binning kernels on the bench's case -> gpurun_out/pmc_bin100m_sq/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bin100m_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=2
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/a -o a --output-format csv -- python3 $R/tools/exp_bin100m.py > $O/a.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA -d $O/b -o b --output-format csv -- python3 $R/tools/exp_bin100m.py > $O/b.log 2>&1
