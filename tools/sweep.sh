#!/bin/bash
# GPU-side experiment: phase ablation of k_pe_tiles and the locus-sort / LDS-aggregation switches
export VS_EXPERIMENT=timing  # the switches below exist only in experiment mode (VsTuning)
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["pe_stats"]["slow_pairs_per_step"], d["ms_per_step"])'
run() { timeout 300 python bench.py --pairs ${PAIRS:-10000000} --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"; }
echo "== default"; run
if [ "$1" != "short" ]; then
echo "== VS_NO_AGG=1"; VS_NO_AGG=1 run
echo "== VS_NO_SORT=1"; VS_NO_SORT=1 run
echo "== VS_NO_SORT=1 VS_NO_AGG=1"; VS_NO_SORT=1 VS_NO_AGG=1 run
fi
for stop in 1 2 3 4 5; do echo "== debug_stop=$stop"; VS_DEBUG_STOP=$stop run; done
for ept in 32 64 96 128; do for g in 8 16; do echo "== ept=$ept grid_per_cu=$g"; VS_EPT=$ept VS_GRID_PER_CU=$g run; done; done
