#!/bin/bash
# GPU-side, round 3 first contact: parity of the PE path (k_pe_walk on certified graphs, seed kernels elsewhere), then
# bench lines with and without the walk kernel.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_pe_gpu.py -x -q -k "not full_size and not campaign and not many_nodes" 2>&1 | tail -15 | tee gpurun_out/r3_pe_tests.log
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f frac %.3f slow_pairs %d matches %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], r["frac"], d["pe_stats"]["slow_pairs_per_step"], d.get("cpu_baseline",{}).get("gpu_matches_on_sample")))'
for cfg in 2 1; do
  echo "== config $cfg walk"; VS_EXPERIMENT=1 VS_WALK=1 timeout 600 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-seconds 4 --no-extract 2>gpurun_out/r3_bench_c${cfg}.err | tee gpurun_out/r3_bench_c${cfg}.json | python -c "$P" || tail -5 gpurun_out/r3_bench_c${cfg}.err
  echo "== config $cfg seeds (VS_NO_WALK=1)"; timeout 600 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"
done
