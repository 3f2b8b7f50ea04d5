#!/bin/bash
# GPU-side: tiles per workgroup run / tile size re-swept under the path numbering
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export VS_EXPERIMENT=1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("tiles %.3f counters %.3f sort %.3f step %.3f" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], d["ms_per_step"]))'
for c in 2 4; do
  for env in "X=0" "VS_GRID_PER_CU=32" "VS_GRID_PER_CU=64" "VS_GRID_PER_CU=256" "VS_GRID_PER_CU=512" "VS_NO_XCD_MAP=1" "VS_ACC_GRID_PER_CU=16" "VS_ACC_GRID_PER_CU=64" "X=1"; do
    echo -n "config $c [$env]: "; env $env timeout 600 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$P"
  done
done 2>&1 | tee gpurun_out/r3_grid.log
