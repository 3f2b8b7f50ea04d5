cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -40 > gpurun_out/r5_gpu_suite_1.log; tail -5 gpurun_out/r5_gpu_suite_1.log
