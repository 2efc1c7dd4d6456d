# HBM counters of the binning kernels (50M points, 57 windows): two passes, one counter each
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_bin
cd /tmp && export TMPDIR=/tmp
export C4N=50000000 C4S=${1:-57}
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_bin/f -o f --output-format csv -- python3 $R/tools/exp_all.py c4 > $R/gpurun_out/pmc_bin/f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_bin/w -o w --output-format csv -- python3 $R/tools/exp_all.py c4 > $R/gpurun_out/pmc_bin/w.log 2>&1
python3 - <<PY
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for name, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    f = glob.glob(f"{R}/gpurun_out/pmc_bin/{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and ("bin_" in r["Kernel_Name"]):
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(name, k, len(v), "mean KiB", round(sum(v) / len(v)))
PY
