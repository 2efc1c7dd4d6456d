# PMC passes over the fused fit+residual launch 
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ONLY=fused
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pmcA_$1 -o a --output-format csv -- python3 $R/tools/exp_fused.py > $R/gpurun_out/pmcA_$1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC -d $R/gpurun_out/pmcB_$1 -o b --output-format csv -- python3 $R/tools/exp_fused.py > $R/gpurun_out/pmcB_$1.log 2>&1
ls $R/gpurun_out/pmcA_$1 $R/gpurun_out/pmcB_$1
