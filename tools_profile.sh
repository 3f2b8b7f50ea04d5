#!/bin/bash
# GPU-side: kernel trace + HBM counters for the bench command (outputs under gpurun_out/prof_r1)
R="$GRAFT_REPO_ROOT"; OUT="$R/gpurun_out/prof_r1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 5 --warmup 1 --cpu-seconds 0 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 > "$OUT/pmc_l2_bench.json" 2> "$OUT/pmc_l2.err"
find "$OUT" -name "*.csv" | head -40; du -sh "$OUT"
