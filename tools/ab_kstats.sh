#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of the bench step for library variants on ONE box:
#   tools/ab_kstats.sh <config> <name> [<name> ...]     (names: tools/_ab/<name>.so, or "own" for the tree's library)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
cfg="$1"; shift
cp vstrains_amd/libvstrains_hip.so /tmp/_own.so
for v in "$@"; do
  [ "$v" = own ] && cp /tmp/_own.so vstrains_amd/libvstrains_hip.so || cp tools/_ab/$v.so vstrains_amd/libvstrains_hip.so
  OUT=/tmp/kst_$v; rm -rf $OUT
  (cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$R/bench.py" --config $cfg --steps 5 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 --no-extract > /dev/null 2> $OUT.err)
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== [$v] config $cfg"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-58s calls %5s avg %10.1f us  total %9.2f ms" % (r["Name"].split("(")[0].replace("void ", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  cp "$f" gpurun_out/kernel_stats_${v}_config$cfg.csv 2>/dev/null
done 2>&1 | tee -a gpurun_out/ab_kstats.log
cp /tmp/_own.so vstrains_amd/libvstrains_hip.so
