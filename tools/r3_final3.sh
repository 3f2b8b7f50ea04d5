#!/bin/bash
# GPU-side, last call of round 3: the new compile-time shape of configs[3] against the oracle (campaign), its profile and bench line, the whole suite
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
FUZZ_STD=1 python tests/fuzz_pe.py ${T_STD:-150} 61 2>&1 | tail -1 | tee gpurun_out/r3_fuzz_pe_std.log
bash tools/profile.sh r3_c3 3 --no-extract > gpurun_out/prof_r3_c3.log 2>&1
cd "$R/gpurun_out"; d=prof_r3_c3; mkdir -p keep_$d; cp $d/pmc_summary.json $d/trace_bench.json keep_$d/ 2>/dev/null; find $d/trace -name "*kernel_stats.csv" -exec cp {} keep_$d/kernel_stats.csv \; ; rm -rf $d; mv keep_$d $d; cd "$R"
cp gpurun_out/prof_r3_c3/pmc_summary.json profiles/r3/pmc_summary_config3.json
timeout 900 python bench.py --config 3 --steps 10 --warmup 1 > gpurun_out/r3_bench_config3_final.json 2> gpurun_out/r3_bench_config3_final.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_gpu_tests_final.log; grep -E "passed|failed" gpurun_out/r3_gpu_tests_final.log
