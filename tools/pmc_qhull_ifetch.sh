# instruction-fetch and LDS counters of the device replay of Qhull's construction -> gpurun_out/pmc_qhull_i/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_qhull_i
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc_qhull_i/a -o a --output-format csv -- python3 $R/tools/exp_qhull_dev.py 2000 50000 > $R/gpurun_out/pmc_qhull_i/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_LEVEL_LDS -d $R/gpurun_out/pmc_qhull_i/b -o b --output-format csv -- python3 $R/tools/exp_qhull_dev.py 2000 50000 > $R/gpurun_out/pmc_qhull_i/b.log 2>&1
