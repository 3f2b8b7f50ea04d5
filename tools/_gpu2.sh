cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5_pe_tests_2.log; tail -3 gpurun_out/r5_pe_tests_2.log
bash tools/campaign.sh sweep "4 2" "X=0"
