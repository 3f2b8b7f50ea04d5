#!/bin/bash
# Build a variant of libvstrains_hip.so from the CURRENT csrc with extra compiler defines into tools/_ab/<name>.so
# (git-ignored, travels with gpurun): tools/ab_build.sh <name> [-DFOO=1 ...].  For A/B sweeps on ONE box (tools/ab_sweep.sh).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; name="$1"; shift
W="$R/tools/_ab/build_$name"; rm -rf "$W"; mkdir -p "$W/vs/csrc" "$W/include"
cp "$R"/vstrains_amd/csrc/*.hip "$R"/vstrains_amd/csrc/*.h "$R"/vstrains_amd/csrc/*.cpp "$R"/vstrains_amd/csrc/Makefile "$W/vs/csrc/"
cp "$R"/include/*.h "$W/include/"
# (the Makefile names ../../include relative to csrc)
mkdir -p "$W/vs/include"; cp "$R"/include/*.h "$W/vs/include/" 2>/dev/null || true
( cd "$W/vs/csrc" && make -s -j6 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off $*" )
cp "$W/vs/libvstrains_hip.so" "$R/tools/_ab/$name.so"; rm -rf "$W"; ls -la "$R/tools/_ab/$name.so"
