#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01g
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/r01g/bench.json 2> $R/gpurun_out/r01g/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01g/trace -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01g/bench_prof.log 2>&1
cat $R/gpurun_out/r01g/bench.json
