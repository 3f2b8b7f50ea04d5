#!/bin/bash
# GPU-side: bench lines (mapping kernel / accumulate / sort / slow / step) for the configs given in $CFGS (default "2"),
# with and without the walk kernel
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f frac %.3f slow_pairs %d matches %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], r["frac"], d["pe_stats"]["slow_pairs_per_step"], d.get("cpu_baseline",{}).get("gpu_matches_on_sample")))'
for cfg in ${CFGS:-2}; do
  echo "== config $cfg walk"; VS_EXPERIMENT=1 VS_WALK=1 timeout 900 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-seconds ${CPUS:-3} --no-extract 2>/tmp/err.txt | python -c "$P" || tail -5 /tmp/err.txt
  if [ -z "$NOSEED" ]; then echo "== config $cfg seeds"; timeout 900 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"; fi
done
