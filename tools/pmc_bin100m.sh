# HBM bytes and times of the binning kernels on the bench's case (100 M points, 2000 windows, random z order):
# a pass per counter, then the kernel trace -> gpurun_out/pmc_bin100m/summary.txt
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bin100m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=2
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/tools/exp_bin100m.py > $O/f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/tools/exp_bin100m.py > $O/w.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 $R/tools/exp_bin100m.py > $O/t.log 2>&1
python3 - <<PY > $O/summary.txt
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bin100m"
n = 100_000_000
val = {}
for name, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "bin" in r["Kernel_Name"] and "pcfit" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    val[name] = {k: sum(v) / len(v) for k, v in agg.items()}
ns = {}
f = glob.glob(f"{O}/t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bin" in r["Name"] and "pcfit" in r["Name"]:
        ns[r["Name"].split("(")[0][:70]] = float(r["AverageNs"])
# rocprofv3 gives both counters in KiB; on gfx950 FETCH_SIZE holds half the bytes of wide streaming reads (128-byte
# requests tallied as 64), so it is doubled; WRITE_SIZE is taken as it is.  (For narrow strided reads the doubling
# may overstate: it is an upper bound there.)
print("kernel; mean launch us; HBM read MB; HBM written MB; bytes per point; TB/s")
tot_r = tot_w = tot_t = 0.0
for k in sorted(ns, key=lambda k: -ns[k]):
    rd = val["FETCH_SIZE"].get(k, 0.0) * 1024 * 2
    wr = val["WRITE_SIZE"].get(k, 0.0) * 1024
    tot_r += rd; tot_w += wr; tot_t += ns[k]
    print(f"{k}; {ns[k] / 1e3:.1f}; {rd / 1e6:.1f}; {wr / 1e6:.1f}; {(rd + wr) / n:.2f}; {(rd + wr) / ns[k] / 1e3:.2f}")
print(f"all binning kernels; {tot_t / 1e3:.1f}; {tot_r / 1e6:.1f}; {tot_w / 1e6:.1f}; {(tot_r + tot_w) / n:.2f}; {(tot_r + tot_w) / tot_t / 1e3:.2f}")
PY
cat $O/summary.txt; tail -2 $O/t.log
