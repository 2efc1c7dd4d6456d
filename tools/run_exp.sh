#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/exp_resident.py 100000000 2000 2>&1 | tail -1 || exit 1
timeout -k 10 500 python3 tools/exp_resident.py 50000000 1000 2>&1 | tail -1 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
