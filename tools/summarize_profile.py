"""Condense the rocprofv3 outputs of tools/profile_bench.sh into profiles/<tag>_*.csv and profiles/traffic.json."""
import csv, glob, json, os, sys, collections
tag, name = sys.argv[1], sys.argv[2]          # gpurun_out/<tag>, profiles/<name>_...
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(root, "profiles", f"{name}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    for r in rows[:6]:
        print(r)
out = [["counter", "kernel", "dispatches", "mean_KiB", "min_KiB", "max_KiB"]]
per = {}
for cname, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.append([cname, k, len(v), sum(v) / len(v), min(v), max(v)])
        per.setdefault(k, {})[cname] = sum(v) / len(v)
with open(os.path.join(root, "profiles", f"{name}_pmc_hbm.csv"), "w", newline="") as f:
    csv.writer(f).writerows(out)
traffic = {"note": "HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters, separate --pmc passes; gfx950 "
                   "FETCH_SIZE reports half of a streaming read), bench.py --steps 3, 10000 slices x 8192 points",
           "algorithmic_bytes_per_launch": 10000 * 8192 * 16}
for k, v in per.items():
    if "fit_residual_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        traffic["fit_residual_kernel_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        traffic["fit_residual_kernel_fetch_KiB"] = v["FETCH_SIZE"]
        traffic["fit_residual_kernel_write_KiB"] = v["WRITE_SIZE"]
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(traffic)
