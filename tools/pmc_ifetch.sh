# instruction-fetch counters of the fused fit+residual launch -> gpurun_out/pmcI_$1
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ONLY=fused
timeout -s KILL 90 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcI_$1 -o i --output-format csv -- python3 $R/tools/exp_fused.py > $R/gpurun_out/pmcI_$1.log 2>&1
