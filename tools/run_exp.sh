#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -u tests/fuzz_parity.py 200 31000 > gpurun_out/fuzz_bin.txt 2>&1 || { tail -20 gpurun_out/fuzz_bin.txt; exit 1; }
tail -1 gpurun_out/fuzz_bin.txt
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bin or scan" 2>&1 | tail -2
