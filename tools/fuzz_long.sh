#!/bin/bash
# GPU-side: the randomized campaigns at length (round-end evidence): tools/fuzz_long.sh [T_PE T_STD T_ROWS T_GRAPH]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
F='^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl\|amdgpu.ids'
python tests/fuzz_pe.py ${1:-900} 71 2>&1 | grep -v "$F" | tail -3 | tee gpurun_out/fuzz_long_pe_boundaries.log
FUZZ_STD=1 python tests/fuzz_pe.py ${2:-900} 72 2>&1 | grep -v "$F" | tail -3 | tee gpurun_out/fuzz_long_pe_std.log
FUZZ_ROWS=1 python tests/fuzz_pe.py ${3:-300} 73 2>&1 | grep -v "$F" | tail -3 | tee gpurun_out/fuzz_long_pe_rows.log
python tests/fuzz_graph.py ${4:-600} 74 2>&1 | grep -v "$F" | tail -3 | tee gpurun_out/fuzz_long_graph.log
