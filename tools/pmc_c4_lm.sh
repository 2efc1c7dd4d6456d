# counters of the Kasa + LM launch on config-4-shaped slices: SQ pass, then the two HBM passes, and a kernel trace
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c4lm
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES -d $R/gpurun_out/c4lm/sq -o a --output-format csv -- python3 $R/tools/exp_c4_lm.py > $R/gpurun_out/c4lm/sq.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/c4lm/f -o f --output-format csv -- python3 $R/tools/exp_c4_lm.py > $R/gpurun_out/c4lm/f.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4lm/t -o t --output-format csv -- python3 $R/tools/exp_c4_lm.py > $R/gpurun_out/c4lm/t.log 2>&1 &&
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/c4lm/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fit_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(f"{k:24s} {len(agg[k])} {sum(agg[k])/len(agg[k]):16.0f}")
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/c4lm/t/*kernel_stats.csv")[0])):
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, "ms")
PY
