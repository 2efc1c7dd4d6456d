# scratch driver for timing experiments on the GPU box
set -e
cd $GRAFT_REPO_ROOT
for S in 57 2000; do
    echo "== S=$S"
    C4N=50000000 C4S=$S timeout -k 10 120 python3 -u tools/exp_all.py c4 2>&1 | grep -E "bin scatter"
    echo "== S=$S stores folded into a hot region"
    PCFIT_BIN_EXP_FOLD=1 C4N=50000000 C4S=$S timeout -k 10 120 python3 -u tools/exp_all.py c4 2>&1 | grep -E "bin scatter"
done
