"""Counters of a rocprofv3 --pmc run, one line per kernel and counter (mean over the dispatches):
python tools/pmc_table.py gpurun_out/<dir> [points per launch, for per-point figures]"""
import collections, csv, glob, sys
per = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
last = None
for (k, c), v in sorted(agg.items()):
    if k != last:
        print(k)
        last = k
    m = sum(v) / len(v)
    print(f"    {c:26s} n={len(v):3d} mean={m:14.5g}" + (f"   x64/point={m * 64 / per:8.2f}" if per and c.startswith("SQ_INSTS") else ""))
