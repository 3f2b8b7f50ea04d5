#!/bin/bash
# GPU-side: parity tests for the PE kernels, then timing lines (map ms, accumulate ms, sort ms, slow ms, step ms)
# QUICK_ENVS="A=1 B=2": one extra timing line per listed setting
export VS_EXPERIMENT=timing  # the switches below exist only in experiment mode (VsTuning)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_pe_gpu.py -x -q 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"])'
run() { timeout 300 python bench.py --pairs ${PAIRS:-10000000} --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"; }
echo "== default"; run
for v in $QUICK_ENVS; do echo "== $v"; export "$v"; run; unset "${v%%=*}"; done
if [ "$1" = "full" ]; then
echo "== VS_NO_AGG=1"; VS_NO_AGG=1 run
echo "== VS_NO_SORT=1"; VS_NO_SORT=1 run
for stop in 1 2 3 4; do echo "== debug_stop=$stop"; VS_DEBUG_STOP=$stop run; done
for ept in 32 64 96 128; do echo "== ept=$ept"; VS_EPT=$ept run; done
fi
