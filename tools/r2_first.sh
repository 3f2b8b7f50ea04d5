#!/bin/bash
# GPU-side: round-2 first pass -- GPU tests, then one bench line per config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2_tests.log
for c in 2 1 3 4; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/r2_bench_c$c.json 2> gpurun_out/r2_bench_c$c.err
  tail -c 600 gpurun_out/r2_bench_c$c.err
done
cat gpurun_out/r2_tests.log
