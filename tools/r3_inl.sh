#!/bin/bash
# GPU-side: parity of the PE path, then k_pe_tiles with 32-byte postings (text around the seed inside the record) against
# the 16-byte ones, per config in $CFGS
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_pe_gpu.py -x -q -k "not full_size and not many_nodes and not two_ranks and not campaign" 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f frac %.3f frac_step %.3f slow_pairs %d matches %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], r["frac"], r["frac_step"], d["pe_stats"]["slow_pairs_per_step"], d.get("cpu_baseline",{}).get("gpu_matches_on_sample")))'
for cfg in ${CFGS:-2}; do
  echo "== config $cfg VS_INLINE=1"; VS_EXPERIMENT=1 VS_INLINE=1 timeout 900 python bench.py --config $cfg --steps 10 --warmup 1 --cpu-seconds 3 --no-extract 2>/tmp/err.txt | python -c "$P" || tail -5 /tmp/err.txt
  echo "== config $cfg default"; timeout 900 python bench.py --config $cfg --steps 10 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"
done
