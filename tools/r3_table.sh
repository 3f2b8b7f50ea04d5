#!/bin/bash
# GPU-side: seed table sized by the distinct seeds (VS_TABLE_SHIFT: slots >= distinct << shift; 8 = as large as the old sizing by positions)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export VS_EXPERIMENT=1
python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3_table_tests.log; tail -3 gpurun_out/r3_table_tests.log
for c in ${TABLE_CONFIGS:-2 4 3 1}; do
  for sh in 8 3 2 1; do
    VS_TABLE_SHIFT=$sh timeout 600 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('config $c shift $sh: step %.3f ms  tiles %.3f  counters %.3f  sort %.3f  overflow %.3f  slots %d distinct %d' % (d['ms_per_step'], r['kernel_ms_avg'], r['accumulate_ms_avg'], r['locus_sort_ms_avg'], r['slow_kernel_ms_avg'], d['config']['index']['slots'], d['config']['index']['distinct_seeds']))"
  done
done 2>&1 | tee gpurun_out/r3_table.log
