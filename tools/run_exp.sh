#!/bin/bash
mkdir -p gpurun_out
C4N=50000000 C4S=57 timeout -k 10 300 python3 tools/exp_all.py c4 2>&1 | grep "split kasa+lm\|sweeps" || exit 1
timeout -k 10 300 python3 tools/exp_analyzer.py 2>&1 | tail -12 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
