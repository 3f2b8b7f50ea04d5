#!/bin/bash
# GPU-side: the randomized campaigns at length (PE: kernel boundaries incl. the 63-base seeds of k >= 95, the compile-time
# tile shapes, VS_WALK / VS_INLINE variants; graph stages: device against checker)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tests/fuzz_pe.py ${T1:-900} 31 2>&1 | tail -3 | tee gpurun_out/r3_fuzz_pe_boundaries.log
FUZZ_STD=1 python tests/fuzz_pe.py ${T2:-300} 32 2>&1 | tail -3 | tee gpurun_out/r3_fuzz_pe_std.log
FUZZ_WALK=1 python tests/fuzz_pe.py ${T3:-300} 33 2>&1 | tail -3 | tee gpurun_out/r3_fuzz_pe_walk.log
python tests/fuzz_graph.py ${T4:-600} 34 2>&1 | tail -3 | tee gpurun_out/r3_fuzz_graph.log
