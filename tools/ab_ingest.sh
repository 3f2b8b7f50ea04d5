#!/bin/bash
# A/B of the FASTQ ingest line of bench.py (fastq_ingest) for library variants on ONE box: tools/ab_ingest.sh <name> ... ("own" = the tree's)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
cp vstrains_amd/libvstrains_hip.so /tmp/_own.so
for rep in 1 2; do for v in "$@"; do
  [ "$v" = own ] && cp /tmp/_own.so vstrains_amd/libvstrains_hip.so || cp tools/_ab/$v.so vstrains_amd/libvstrains_hip.so
  echo -n "[$v] "; python bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d["fastq_ingest"]; print("open %.4f pack %.4f pairs/s %.3g" % (f["open_index_s"], f["pack_upload_count_s"], f["pairs_per_s"]))'
done; done 2>&1 | tee -a gpurun_out/ab_ingest.log
cp /tmp/_own.so vstrains_amd/libvstrains_hip.so
