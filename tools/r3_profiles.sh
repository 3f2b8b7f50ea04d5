#!/bin/bash
# GPU-side: rocprofv3 kernel-trace stats + HBM counters per config (outputs under gpurun_out/prof_r3_c<i>), SQ counters for
# configs[2] and configs[3], and the untraced bench lines of HEAD
R="$GRAFT_REPO_ROOT"
mkdir -p "$R/gpurun_out"
for c in ${PROF_CONFIGS:-2 1 3 4 0}; do
  bash "$R/tools/profile.sh" r3_c$c $c --no-extract > "$R/gpurun_out/prof_r3_c$c.log" 2>&1
  tail -2 "$R/gpurun_out/prof_r3_c$c.log"
done
for c in 2 3; do
  CFG=$c bash "$R/tools/pmc.sh" r3_c$c > "$R/gpurun_out/pmc_r3_c$c.log" 2>&1
  tail -4 "$R/gpurun_out/pmc_r3_c$c.log"
done
cd "$R/gpurun_out"
for c in ${PROF_CONFIGS:-2 1 3 4 0}; do
  d=prof_r3_c$c
  mkdir -p keep_$d
  cp $d/pmc_summary.json $d/trace_bench.json keep_$d/ 2>/dev/null
  find $d/trace -name "*kernel_stats.csv" -exec cp {} keep_$d/kernel_stats.csv \;
  rm -rf $d; mv keep_$d $d
done
for c in 2 3; do mkdir -p keep_pmc; cp pmc_r3_c$c/summary.json keep_pmc/ 2>/dev/null; rm -rf pmc_r3_c$c; mv keep_pmc pmc_r3_c$c; done
cd "$R"
for c in 2 1 0; do timeout 900 python bench.py --config $c --steps 20 --warmup 2 > gpurun_out/r3_bench_config${c}_final.json 2> gpurun_out/r3_bench_config${c}_final.err; tail -c 300 gpurun_out/r3_bench_config${c}_final.err; done
for c in 3 4; do timeout 1200 python bench.py --config $c --steps 10 --warmup 1 > gpurun_out/r3_bench_config${c}_final.json 2> gpurun_out/r3_bench_config${c}_final.err; tail -c 300 gpurun_out/r3_bench_config${c}_final.err; done
du -sh gpurun_out/prof_r3_c* gpurun_out/pmc_r3_c*
