cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -6 | tee gpurun_out/r5_pe_tests_adapt.log
FUZZ_STD=1 timeout 400 python tests/fuzz_pe.py 240 51 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -4 | tee gpurun_out/r5_fuzz_std_adapt.log
timeout 300 python tests/fuzz_pe.py 150 52 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -4 | tee gpurun_out/r5_fuzz_boundaries_adapt.log
bash tools/campaign.sh sweep "2 3 4 1" "X=default" "VS_ADAPT_GRID=1" "VS_ADAPT_GRID=0"
