set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -3 gpurun_out/t_gpu.log
timeout -k 10 300 python -u tools/exp_hull.py 2>&1 | grep -v amdgpu
