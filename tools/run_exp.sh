#!/bin/bash
timeout -k 10 300 python3 bench.py --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step']); [print(k) for k in d['kernels']]" || exit 1
