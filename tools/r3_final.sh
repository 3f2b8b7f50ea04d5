#!/bin/bash
# GPU-side, end of round 3: the whole GPU suite, the profile of configs[4] (its sort / overflow / counter kernels changed
# last), the bench lines of HEAD for every config, the extract leg at configs[3] / [4]
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3_gpu_tests_final.log; tail -4 gpurun_out/r3_gpu_tests_final.log
bash tools/profile.sh r3_c4 4 --no-extract > gpurun_out/prof_r3_c4.log 2>&1; tail -2 gpurun_out/prof_r3_c4.log
cd "$R/gpurun_out"; d=prof_r3_c4; mkdir -p keep_$d; cp $d/pmc_summary.json $d/trace_bench.json keep_$d/ 2>/dev/null; find $d/trace -name "*kernel_stats.csv" -exec cp {} keep_$d/kernel_stats.csv \; ; rm -rf $d; mv keep_$d $d; cd "$R"
for c in 2 1 0; do timeout 900 python bench.py --config $c --steps 20 --warmup 2 > gpurun_out/r3_bench_config${c}_final.json 2> gpurun_out/r3_bench_config${c}_final.err; tail -c 200 gpurun_out/r3_bench_config${c}_final.err; done
for c in 3 4; do timeout 1200 python bench.py --config $c --steps 10 --warmup 1 > gpurun_out/r3_bench_config${c}_final.json 2> gpurun_out/r3_bench_config${c}_final.err; tail -c 200 gpurun_out/r3_bench_config${c}_final.err; done
for c in 3 4; do timeout 1500 python bench.py --config $c --steps 2 --warmup 1 --extract --cpu-seconds 0 --ingest-pairs 0 > gpurun_out/r3_bench_config${c}_with_extract.json 2>/dev/null; done
