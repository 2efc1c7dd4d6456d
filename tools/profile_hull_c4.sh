# kernel trace of the hybrid chain on config-4-like slices (2000 x 50000 by default) -> gpurun_out/hull_c4/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hull_c4
cd /tmp && export TMPDIR=/tmp
PHASES=0 timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hull_c4 -o h --output-format csv -- python3 $R/tools/exp_hull_c4.py ${1:-2000} ${2:-50000} > $R/gpurun_out/hull_c4/run.log 2>&1
cut -c 1-90,200-400 $R/gpurun_out/hull_c4/h_kernel_stats.csv | head -12
