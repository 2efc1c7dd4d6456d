#!/bin/bash
for a in 0 1; do
echo "PCFIT_LM_ADAPTIVE=$a"
PCFIT_LM_ADAPTIVE=$a timeout -k 10 300 python3 tools/exp_resident.py 2>&1 | tail -1 || exit 1
PCFIT_LM_ADAPTIVE=$a timeout -k 10 300 python3 tools/exp_resident.py 20000000 57 2>&1 | tail -1 || exit 1
PCFIT_LM_ADAPTIVE=$a timeout -k 10 300 python3 tools/exp_resident.py 50000000 200 2>&1 | tail -1 || exit 1
done
