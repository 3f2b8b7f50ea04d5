#!/bin/bash
# GPU-side: grouped cell table (AccGrp) -- parity tests, then the bench per config against the old shapes and fill limits
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export VS_EXPERIMENT=1
python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_grp_tests.log; tail -3 gpurun_out/r3_grp_tests.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("tiles %.3f counters %.3f sort %.3f overflow %.3f step %.3f match %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d["cpu_baseline"].get("gpu_matches_on_sample")))'
for c in 4 2 3 1; do
  for env in "X=0" "VS_ACC_FILL=6" "VS_ACC_FILL=25" "VS_ACC_FILL=50" "VS_ACC_FILL=100" "VS_ACC_MERGE=1" "VS_ACC_WIDE=2" "VS_ACC_WIDE=3"; do
    echo -n "config $c [$env]: "; env $env timeout 600 python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 3 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$P"
  done
done 2>&1 | tee gpurun_out/r3_grp.log
