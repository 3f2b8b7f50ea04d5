#!/bin/bash
# GPU-side: compile-time shape of the long-window kernel for k = 127, 2 x 250 (k_pe_tiles<2, 16, 3>) against the run-time shape
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export VS_EXPERIMENT=1
python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "127 or long or config3 or golden or repeated or random_graphs or digest" 2>&1 | tail -5 > gpurun_out/r3_std2_tests.log; grep -E "passed|failed" gpurun_out/r3_std2_tests.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%s tiles %.3f counters %.3f sort %.3f step %.3f match %s" % (r["kernel"], r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], d["ms_per_step"], d["cpu_baseline"].get("gpu_matches_on_sample")))'
for env in "VS_NO_STD=1" "X=0" "VS_NO_STD=1" "X=0"; do
  echo -n "config 3 [$env]: "; env $env timeout 600 python bench.py --config 3 --steps 10 --warmup 2 --cpu-seconds 3 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$P"
done 2>&1 | tee gpurun_out/r3_std2.log
FUZZ_STD=1 python tests/fuzz_pe.py 60 55 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r3_std2.log
