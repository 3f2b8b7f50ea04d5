#!/bin/bash
# GPU-side: SQ / TCP / TCC counter passes over the bench command (outputs under gpurun_out/pmc_$1)
R="$GRAFT_REPO_ROOT"; TAG="${1:-x}"; OUT="$R/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config ${CFG:-2} --steps 2 --warmup 1 --cpu-seconds 0 --no-extract"
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $B > "$OUT/$name.json" 2> "$OUT/$name.err"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    disp = collections.Counter()
    for (kn, cn), (v, n) in acc.items():
        res.setdefault(kn, {})[cn] = v
    # dispatch counts: rows per (kernel,counter) may span dimensions; keep raw sums plus row counts
    for (kn, cn), (v, n) in acc.items():
        res[kn][cn + "#rows"] = n
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for kn, d in res.items():
    if "k_pe" in kn or "k_locus" in kn:
        print(kn, {k: round(v / max(d.get(k + "#rows", 1), 1)) for k, v in d.items() if not k.endswith("#rows")})
PY
