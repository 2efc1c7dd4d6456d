#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c4r
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4r/trace -o s --output-format csv -- python3 $R/tools/exp_resident.py 100000000 2000 > $R/gpurun_out/c4r/log.txt 2>&1
tail -1 $R/gpurun_out/c4r/log.txt
