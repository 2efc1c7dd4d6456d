# scratch driver for timing experiments on the GPU box
set -e
cd $GRAFT_REPO_ROOT
for S in 57 200; do
  for PAD in 0 5000 10000 20000 40000 80000; do
    echo "== S=$S pad=$PAD"
    PCFIT_BIN_LDS_PAD=$PAD C4N=50000000 C4S=$S timeout -k 10 120 python3 -u tools/exp_all.py c4 2>&1 | grep -E "bin scatter"
  done
done
