# kernel trace of ten analyses of a resident scan (default 100 M points, 2000 windows) -> gpurun_out/resident/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/resident
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/resident -o r --output-format csv -- python3 $R/tools/exp_resident.py ${1:-100000000} ${2:-2000} ${3:-0.0} ${4:-qhull} > $R/gpurun_out/resident/run.log 2>&1
tail -2 $R/gpurun_out/resident/run.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/resident/r_kernel_stats.csv")))
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / 11.0
    if "bin" in r["Name"]:
        tot += ms
    print(f"{ms:8.3f} ms/analysis  {int(r['Calls']):5d} calls  {r['Name'][:100]}")
print(f"binning kernels: {tot:.3f} ms per analysis")
PY
