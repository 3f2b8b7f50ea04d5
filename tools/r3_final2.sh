#!/bin/bash
# GPU-side, end of round 3 (after the path numbering and the ACC_SEG cell table): whole GPU suite, randomized campaigns,
# profiles of every config (kernel stats + HBM / L2 counters; SQ counters for configs 2 and 3), bench lines of HEAD
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3_gpu_tests_final.log; tail -3 gpurun_out/r3_gpu_tests_final.log
T1=420 T2=200 T3=150 T4=300 bash tools/r3_fuzz.sh
bash tools/r3_profiles.sh
for c in 3 4; do timeout 1500 python bench.py --config $c --steps 2 --warmup 1 --extract --cpu-seconds 0 --ingest-pairs 0 > gpurun_out/r3_bench_config${c}_with_extract.json 2>/dev/null; done
