#!/bin/bash
# the round-end sequence: GPU tests, smoke, bench
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 400 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -5 gpurun_out/bench_final.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
