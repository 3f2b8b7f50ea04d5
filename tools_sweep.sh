#!/bin/bash
# GPU-side experiment: phase ablation and tile-size sweep of k_pe_tiles (timing only)
cd "$GRAFT_REPO_ROOT"
for stop in 1 2 3 4 5 0; do
  echo "== debug_stop=$stop"; VS_DEBUG_STOP=$stop timeout 200 python bench.py --pairs 4000000 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['slow_kernel_ms_avg'], d['pe_stats']['slow_pairs_per_step'])"
done
for ept in 32 64 96 128; do for g in 4 8 16; do
  echo "== ept=$ept grid_per_cu=$g"; VS_EPT=$ept VS_GRID_PER_CU=$g timeout 200 python bench.py --pairs 4000000 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['slow_kernel_ms_avg'], d['pe_stats']['slow_pairs_per_step'])"
done; done
