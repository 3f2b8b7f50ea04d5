cd $GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -3
FUZZ_STD=1 timeout 300 python tests/fuzz_pe.py 120 61 2>&1 | grep -v "$F" | tail -2 | cut -c1-200
bash tools/campaign.sh sweep "4 2" "X=mid"
