# rocprofv3 kernel statistics of the timing survey (tools/exp_all.py) -> gpurun_out/$1/
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-survey}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
C4N=50000000 C4S=57 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o s --output-format csv -- python3 $R/tools/exp_all.py c4 c3 c1 > $R/gpurun_out/$TAG/survey.log 2>&1
cat $R/gpurun_out/$TAG/survey.log | tail -30
