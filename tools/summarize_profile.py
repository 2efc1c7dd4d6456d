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
# the timed launches alone: bench.py runs the kernel 64 times before its warmup steps (the clock of an idle GPU comes
# up over the first ~40), so the mean over all launches in the stats file lies a little above the mean of the last 200
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "fit_residual_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if len(dur) >= 200:
        with open(os.path.join(root, "profiles", f"{name}_kernel_trace_fused_launches.txt"), "w") as f:
            f.write(f"# fit_residual_kernel launches of `bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-side` "
                    f"under rocprofv3 --kernel-trace, in order: {len(dur)} launches (64 to settle the clock, 20 warmup, 200 timed)\n")
            f.write(f"mean of all launches: {sum(dur) / len(dur):.1f} us\n")
            f.write(f"mean of launches 2 to 21 (from idle): {sum(dur[1:21]) / 20:.1f} us\n")
            f.write(f"mean of the last 200 (the timed steps): {sum(dur[-200:]) / 200:.1f} us\n")
            f.write("every 10th launch, us: " + " ".join(f"{v:.0f}" for v in dur[::10]) + "\n")
        print(open(os.path.join(root, "profiles", f"{name}_kernel_trace_fused_launches.txt")).read())
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
sys.path.insert(0, root)
from bench import kernel_sources_sha1
traffic = {"source": [f"profiles/{name}_pmc_hbm.csv", f"profiles/{name}_bench_pmc_sq.csv"],
           "kernel_sources_sha1": kernel_sources_sha1(),
           "note": "HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters, separate --pmc passes; gfx950 "
                   "FETCH_SIZE reports half of a streaming read), bench.py --steps 3 --warmup 1 (+ 64 launches before them), 10000 slices x 8192 points",
           "algorithmic_bytes_per_launch": 10000 * 8192 * 16}
for k, v in per.items():
    if "fit_residual_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        traffic["fit_residual_kernel_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        traffic["fit_residual_kernel_fetch_KiB"] = v["FETCH_SIZE"]
        traffic["fit_residual_kernel_write_KiB"] = v["WRITE_SIZE"]
# the SQ passes of the same run (tools/profile_bench.sh): one line per pass, counter and kernel, and the share of the
# SIMDs' cycles in which the fused kernel issued a vector instruction
sq = [["pass", "counter", "kernel", "dispatches", "mean"]]
fused = {}
for d in ("a", "b", "c"):
    for f in glob.glob(os.path.join(src, "sq_" + d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Counter_Name"], r["Kernel_Name"][:60])].append(float(r["Counter_Value"]))
        for (c, k), v in sorted(agg.items()):
            sq.append([d, c, k, len(v), sum(v) / len(v)])
            if "fit_residual_kernel" in k:
                fused[c] = sum(v) / len(v)
if len(sq) > 1:
    with open(os.path.join(root, "profiles", f"{name}_bench_pmc_sq.csv"), "w", newline="") as f:
        csv.writer(f).writerows(sq)
if "SQ_ACTIVE_INST_VALU" in fused and "SQ_BUSY_CYCLES" in fused:
    # SQ_ACTIVE_INST_VALU counts quad-cycles of the SIMDs, SQ_BUSY_CYCLES the kernel's cycles summed over 32 shader
    # engines; 1024 SIMDs: share = 4 * ACTIVE / (BUSY / 32 * 1024)
    traffic["fit_residual_kernel_valu_issue_frac"] = fused["SQ_ACTIVE_INST_VALU"] / (8.0 * fused["SQ_BUSY_CYCLES"])
    traffic["fit_residual_kernel_valu_insts_per_point"] = fused.get("SQ_INSTS_VALU", 0.0) * 64 / (10000 * 8192)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    open(os.path.join(root, "profiles", f"{name}_bench.json"), "w").write(open(bj).read())
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(traffic)
