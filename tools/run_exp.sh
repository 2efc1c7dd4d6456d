#!/bin/bash
# scratch: bench with and without the exchange, one rank
mkdir -p gpurun_out
for m in 0 1; do
  echo "== force_dist=$m"
  PCFIT_BENCH_DEBUG=1 PCFIT_BENCH_FORCE_DIST=$m timeout -k 10 200 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ex_$m.json 2> gpurun_out/ex_$m.err || { tail -5 gpurun_out/ex_$m.err; exit 1; }; grep "\[bench\]" gpurun_out/ex_$m.err
done
