cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off"
cp vstrains_amd/libvstrains_hip.so /tmp/lib_base.so
timeout 1200 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r5_pe_tests_adapt.log
bash tools/campaign.sh sweep "2 3 4" "X=adapt"
for c in 2 3 4; do echo "postings config $c"; env VS_EXPERIMENT=1 VS_DEBUG_POSTINGS=1 VS_NO_STD=1 timeout 600 python bench.py --config $c --steps 1 --warmup 0 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>&1 | grep "postings expanded" | tail -1; done
touch vstrains_amd/csrc/vs_pe.hip
make -s -C vstrains_amd/csrc CXXFLAGS="$BASE -DVS_ADAPT=0" 2>&1 | grep -i "error" | head -3
bash tools/campaign.sh sweep "2 3 4" "X=noadapt"
cp /tmp/lib_base.so vstrains_amd/libvstrains_hip.so
