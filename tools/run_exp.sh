set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -3 gpurun_out/t_gpu.log
timeout -k 10 120 python -u tools/exp_fused.py 2>&1 | grep "fused\|kasa only\|max_iter" > gpurun_out/exp_fused.log
cat gpurun_out/exp_fused.log
