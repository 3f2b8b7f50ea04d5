#!/bin/bash
# GPU-side: extract leg with the deferred stage-file writer (three runs), configs[4] counters with / without list merging
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do python bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extract', d['strain_extract_s'], d['strain_extract']['stages'])"; done
for i in 1; do python bench.py --config 1 --steps 3 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extract c1', d['strain_extract_s'])"; done
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "map %.3f acc %.3f sort %.3f slow %.3f step %.3f" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"]))'
echo "== config 4 default"; timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"
echo "== config 4 VS_ACC_MERGE=1"; VS_EXPERIMENT=1 VS_ACC_MERGE=1 timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 0 --no-extract 2>/dev/null | python -c "$P"
