#!/bin/bash
mkdir -p gpurun_out
C4N=50000000 C4S=57 timeout -k 10 300 python3 tools/exp_all.py c4 2>&1 | grep "bin c"
C4N=100000000 C4S=2000 timeout -k 10 300 python3 tools/exp_all.py c4 2>&1 | grep "bin c"
timeout -k 10 600 python3 -u tests/fuzz_parity.py 100 9000 > gpurun_out/fuzz_bin.txt 2>&1 || { tail -20 gpurun_out/fuzz_bin.txt; exit 1; }
tail -1 gpurun_out/fuzz_bin.txt
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bin" 2>&1 | tail -2
