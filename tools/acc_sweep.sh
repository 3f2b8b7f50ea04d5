#!/bin/bash
# GPU-side: rebuild vs_pe.o with other compile-time constants and time the step (experiments)
#   tools/acc_sweep.sh "-DLOCUS_WGS=2048u" "-DACC_BITS=13" ...      (SWEEP_ENVS="A=1 B=2" adds settings per build)
export VS_EXPERIMENT=timing  # the switches below exist only in experiment mode (VsTuning)
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"])'
run() { timeout 300 python bench.py --config ${CFG:-2} --pairs ${PAIRS:-0} --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"; }
for def in "" "$@"; do
  (cd vstrains_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $def -c vs_pe.hip -o vs_pe.o && make -s) || exit 1
  echo "== build '$def'"; run
  for v in $SWEEP_ENVS; do echo "== build '$def' $v"; export "$v"; run; unset "${v%%=*}"; done
done
