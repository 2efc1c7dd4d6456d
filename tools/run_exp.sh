#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
timeout -k 10 900 python3 -u tests/fuzz_parity.py 600 40000 > gpurun_out/fuzz_bin.txt 2>&1 || { tail -20 gpurun_out/fuzz_bin.txt; exit 1; }
tail -1 gpurun_out/fuzz_bin.txt
