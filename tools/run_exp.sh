set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -3 gpurun_out/t_gpu.log
for s in 57 2000; do
echo "== $s windows" >> gpurun_out/exp_c4.log
C4N=50000000 C4S=$s PCFIT_BIN_CHUNKS=8192 timeout -k 10 200 python -u tools/exp_all.py c4 2>&1 | grep "split\|sweeps\|==" >> gpurun_out/exp_c4.log
done
cat gpurun_out/exp_c4.log
