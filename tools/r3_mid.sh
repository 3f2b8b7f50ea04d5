#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_pe_gpu.py -x -q -k "not full_size and not two_ranks" 2>&1 | tail -5
python -m pytest tests/test_configs_gpu.py -x -q -k "config4" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f slow_pairs %d matches %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d["pe_stats"]["slow_pairs_per_step"], d.get("cpu_baseline",{}).get("gpu_matches_on_sample")))'
echo "== config 4"; timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 2 --no-extract 2>/dev/null | python -c "$P"
echo "== config 4 VS_NO_MID=1"; VS_EXPERIMENT=1 VS_NO_MID=1 timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"
echo "== config 2"; timeout 900 python bench.py --config 2 --steps 5 --warmup 1 --cpu-seconds 2 --no-extract 2>/dev/null | python -c "$P"
