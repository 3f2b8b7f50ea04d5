#!/bin/bash
# GPU-side iteration: PE parity tests, then timing lines per config (map ms, accumulate ms, sort ms, slow ms, step ms)
#   ITER_TESTS="tests/test_pe_gpu.py tests/test_configs_gpu.py"  ITER_CONFIGS="2 1 4 3"  QUICK_ENVS="A=1 B=2"
export VS_EXPERIMENT=timing  # the switches below exist only in experiment mode (VsTuning)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T="${ITER_TESTS:-tests/test_pe_gpu.py}"
timeout 1500 python -m pytest $T -x -q 2>&1 | tail -${ITER_TAIL:-12}
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f pairs/s %.3e slow_pairs %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d["value"], d["pe_stats"]["slow_pairs_per_step"]))'
run() { timeout 600 python bench.py --config $1 --steps 3 --warmup 1 --cpu-seconds ${CPU_S:-0} --no-extract $BENCH_ARGS 2>gpurun_out/iter_c$1.err | python -c "$P" || tail -5 gpurun_out/iter_c$1.err; }
for c in ${ITER_CONFIGS:-2}; do
  echo "== config $c"; run $c
  for v in $QUICK_ENVS; do echo "== config $c $v"; export "$v"; run $c; unset "${v%%=*}"; done
done
if [ -n "$ITER_POSTINGS" ]; then VS_NO_STD=1 VS_DEBUG_POSTINGS=1 timeout 600 python bench.py --config 2 --steps 1 --warmup 0 --cpu-seconds 0 --no-extract 2>&1 >/dev/null | grep "postings expanded" | tail -1; fi
