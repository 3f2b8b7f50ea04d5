#!/bin/bash
# GPU-side: rebuild vs_pe.hip with extra compile flags and time the bench line per variant:
#   tools/variant_sweep.sh "<configs>" "<flag set 1>" "<flag set 2>" ...      (the base library is restored at the end)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off"
configs="$1"; shift
cp vstrains_amd/libvstrains_hip.so /tmp/lib_base.so
bash tools/campaign.sh sweep "$configs" "X=base"
for v in "$@"; do
  touch vstrains_amd/csrc/vs_pe.hip
  make -s -C vstrains_amd/csrc CXXFLAGS="$BASE $v" 2>&1 | grep -i "error" | head -3
  bash tools/campaign.sh sweep "$configs" "X=$v"
  if [ -n "$VARIANT_TESTS" ]; then timeout 900 python -m pytest tests/test_pe_gpu.py -m gpu -x -q -k "$VARIANT_TESTS" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -2; fi
done
cp /tmp/lib_base.so vstrains_amd/libvstrains_hip.so
