# HBM traffic of the device replay of Qhull's construction (tools/exp_qhull_dev.py) -> gpurun_out/pmc_qhull/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_qhull
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_qhull/f -o f --output-format csv -- python3 $R/tools/exp_qhull_dev.py 2000 50000 > $R/gpurun_out/pmc_qhull/f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_qhull/w -o w --output-format csv -- python3 $R/tools/exp_qhull_dev.py 2000 50000 > $R/gpurun_out/pmc_qhull/w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_qhull/s -o s --output-format csv -- python3 $R/tools/exp_qhull_dev.py 2000 50000 > $R/gpurun_out/pmc_qhull/s.log 2>&1
