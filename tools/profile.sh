#!/bin/bash
# GPU-side: kernel trace + HBM counters for the bench command (outputs under gpurun_out/prof_$1)
#   tools/profile.sh TAG [CONFIG] [extra bench args...]
R="$GRAFT_REPO_ROOT"; TAG="${1:-r2}"; CFG="${2:-2}"; shift; shift
OUT="$R/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --config $CFG --steps 5 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 "$@" > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 1 --cpu-seconds 0 --no-extract "$@" > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 1 --cpu-seconds 0 --no-extract "$@" > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum --output-format csv -d "$OUT/pmc_l2" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 1 --cpu-seconds 0 --no-extract "$@" > "$OUT/pmc_l2_bench.json" 2> "$OUT/pmc_l2.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1].add(row["Dispatch_Id"])
    for (kn, cn), (v, ids) in acc.items():
        res.setdefault(kn, {})[cn] = {"per_dispatch_mean": v / max(len(ids), 1), "dispatches": len(ids)}
try:
    res["_pairs_per_gpu"] = json.loads(open(out + "/pmc_fetch_bench.json").read().strip().splitlines()[-1])["config"]["pairs_per_gpu"]
except Exception as e:
    res["_pairs_per_gpu"] = None
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
PY
find "$OUT" -name "*kernel_stats.csv" | head -3; du -sh "$OUT"
