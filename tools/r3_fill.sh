#!/bin/bash
export VS_EXPERIMENT=timing
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("acc %.3f step %.3f" % (r["accumulate_ms_avg"], d["ms_per_step"]))'
for e in "VS_ACC_FILL=6" "VS_ACC_FILL=12" "VS_ACC_FILL=25" "VS_ACC_FILL=50" "VS_ACC_FILL=90" "VS_ACC_FILL=3" "VS_ACC_GRID_PER_CU=8" "VS_ACC_QUEUE=0"; do
  echo "== $e"; env $e VS_DEBUG_ACC=1 timeout 600 python bench.py --config ${CFG:-4} --steps 2 --warmup 1 --cpu-seconds 0 --no-extract 2>/tmp/err.txt | python -c "$P"; grep "k_pe_accumulate:" /tmp/err.txt | tail -1
done
