cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off"
cp vstrains_amd/libvstrains_hip.so /tmp/lib_base.so
timeout 900 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/campaign.sh sweep "3 2" "X=base"
for v in "-DTILES_WAVES_LONG=4" "-DTILES_WAVES_LONG=6"; do
  touch vstrains_amd/csrc/vs_pe.hip
  make -s -C vstrains_amd/csrc CXXFLAGS="$BASE $v" 2>&1 | grep -i "error" | head -3
  bash tools/campaign.sh sweep "3" "X=$v"
done
cp /tmp/lib_base.so vstrains_amd/libvstrains_hip.so
