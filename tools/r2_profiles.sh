#!/bin/bash
# GPU-side: rocprofv3 kernel-trace stats + HBM counters per config (outputs under gpurun_out/prof_r2_c<i>)
R="$GRAFT_REPO_ROOT"
for c in ${PROF_CONFIGS:-2 1 3 4}; do
  bash "$R/tools/profile.sh" r2_c$c $c --no-extract > "$R/gpurun_out/prof_r2_c$c.log" 2>&1
  tail -2 "$R/gpurun_out/prof_r2_c$c.log"
done
# SQ counters for the headline config
CFG=2 bash "$R/tools/pmc.sh" r2_c2 > "$R/gpurun_out/pmc_r2_c2.log" 2>&1
tail -4 "$R/gpurun_out/pmc_r2_c2.log"
# keep only the summaries (the raw per-dispatch CSVs are large)
cd "$R/gpurun_out"
for c in ${PROF_CONFIGS:-2 1 3 4}; do
  d=prof_r2_c$c
  mkdir -p keep_$d
  cp $d/pmc_summary.json $d/trace_bench.json keep_$d/ 2>/dev/null
  find $d/trace -name "*kernel_stats.csv" -exec cp {} keep_$d/kernel_stats.csv \;
  rm -rf $d; mv keep_$d $d
done
mkdir -p keep_pmc; cp pmc_r2_c2/summary.json keep_pmc/ 2>/dev/null; rm -rf pmc_r2_c2; mv keep_pmc pmc_r2_c2
du -sh prof_r2_c* pmc_r2_c2
