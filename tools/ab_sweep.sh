#!/bin/bash
# A/B on one GPU box: for every tools/_ab/<name>.so named, put it in the product's place, run the per-kernel timing line of
# bench.py for the configs, and restore the tree's own library:  tools/ab_sweep.sh "2 4" r5 cur nodpp
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
configs="$1"; shift
cp vstrains_amd/libvstrains_hip.so /tmp/_own.so
LINE='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("tiles %.3f counters %.3f sort %.3f overflow %.3f step %.3f" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"]))'
for rep in 1 2; do for v in "$@"; do
  cp tools/_ab/$v.so vstrains_amd/libvstrains_hip.so
  for c in $configs; do
    echo -n "[$v] config $c: "
    VS_EXPERIMENT=1 timeout 600 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$LINE"
  done
done; done 2>&1 | tee -a gpurun_out/ab_sweep.log
cp /tmp/_own.so vstrains_amd/libvstrains_hip.so
