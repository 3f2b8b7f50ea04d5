#!/bin/bash
# GPU-side: the index numbered along the graph's paths (Context.build_index renumber, csrc/vs_order_host.cpp):
# whole GPU suite, tools/order_probe.py, bench lines with and without
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3_order_tests.log; tail -4 gpurun_out/r3_order_tests.log
for c in 2 4 3; do
  for rn in 1 0; do
    VS_EXPERIMENT=1 VS_RENUMBER=$rn timeout 900 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 5 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('config $c renumber $rn: step %.3f ms  tiles %.3f  counters %.3f  sort %.3f  overflow %.3f  index %.3f s  matches %s  (%s)' % (d['ms_per_step'], r['kernel_ms_avg'], r['accumulate_ms_avg'], r['locus_sort_ms_avg'], r['slow_kernel_ms_avg'], d['config']['index_build_s'], d['cpu_baseline']['gpu_matches_on_sample'], d['config']['node_numbering']))"
  done
done 2>&1 | tee gpurun_out/r3_order_bench.log
ORDER_FROM_RANDOM=1 python tools/order_probe.py 4 > gpurun_out/order_probe_c4.log 2>&1; tail -14 gpurun_out/order_probe_c4.log
