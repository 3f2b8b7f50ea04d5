#!/bin/bash
export VS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f matches %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d.get("cpu_baseline",{}).get("gpu_matches_on_sample")))'
for cfg in ${CFGS:-4 2 3}; do for e in "VS_REFINE=1" "VS_REFINE=0"; do
  echo "== config $cfg $e"; env $e VS_DEBUG_ACC=1 timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 2 --no-extract 2>/tmp/err.txt | python -c "$P"; grep "k_pe_accumulate:" /tmp/err.txt | tail -1
done; done
