cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off"
cp vstrains_amd/libvstrains_hip.so /tmp/lib_base.so
bash tools/campaign.sh sweep "2" "X=base" "VS_GRID_PER_CU=64" "VS_GRID_PER_CU=256" "VS_GRID_PER_CU=512"
for v in "-DPPT=3" "-DPPT=1" "-DTILES_WAVES=4" "-DPPT=4"; do
  touch vstrains_amd/csrc/vs_pe.hip
  make -s -C vstrains_amd/csrc CXXFLAGS="$BASE $v" 2>&1 | grep -i "error" | head -3
  echo "variant [$v]"
  bash tools/campaign.sh sweep "2 3" "X=$v"
done
cp /tmp/lib_base.so vstrains_amd/libvstrains_hip.so
