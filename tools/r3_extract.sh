#!/bin/bash
# GPU-side: the strain-extract leg after the host-logic shortcuts (fork scan, link-table walk, id validation as set operations)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3_extract_tests.log; tail -2 gpurun_out/r3_extract_tests.log
VS_CHECK_UNTOUCHED=0 python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3_extract_tests_nocheck.log; tail -2 gpurun_out/r3_extract_tests_nocheck.log
python tests/fuzz_graph.py 120 77 2>&1 | tail -2 | tee gpurun_out/r3_extract_fuzz.log
VS_CHECK_UNTOUCHED=0 python tests/fuzz_graph.py 120 78 2>&1 | tail -2 | tee -a gpurun_out/r3_extract_fuzz.log
for c in 2 3 4; do
  python bench.py --config $c --steps 2 --warmup 1 --extract --cpu-seconds 0 --ingest-pairs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $c: strain_extract_s', d['strain_extract_s'], d['strain_extract'].get('stages'), d['strain_extract'].get('strains'))" | tee -a gpurun_out/r3_extract.log
done
