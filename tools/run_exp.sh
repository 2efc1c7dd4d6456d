# scratch driver for timing experiments on the GPU box
set -e
cd $GRAFT_REPO_ROOT
for S in 57 200 2000; do
    echo "== S=$S default stores"
    C4N=50000000 C4S=$S timeout -k 10 120 python3 -u tools/exp_all.py c4 2>&1 | grep -E "bin scatter"
    echo "== S=$S non-temporal loads"
    PCFIT_SO=libpcfit_hip_nt.so C4N=50000000 C4S=$S timeout -k 10 120 python3 -u tools/exp_all.py c4 2>&1 | grep -E "bin scatter"
done
