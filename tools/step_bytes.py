#!/usr/bin/env python3
"""Per-kernel HBM bytes, time and rate of one bench step from the committed rocprofv3 summaries of a round:
profiles/r<N>/pmc_summary_config<i>.json (FETCH_SIZE / WRITE_SIZE per dispatch, separate --pmc passes; bytes = 2 x FETCH_SIZE +
WRITE_SIZE, the guide's gfx950 rule) and kernel_stats_config<i>.csv (--kernel-trace --stats of the same command).
    python tools/step_bytes.py --round 6 --config 4 [--out profiles/r6/step_bytes_config4.json]"""
import argparse, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--round", type=int, default=6); ap.add_argument("--config", type=int, default=4); ap.add_argument("--out", default=None)
a = ap.parse_args()
d = os.path.join(ROOT, "profiles", "r%d" % a.round)
pmc = json.load(open(os.path.join(d, "pmc_summary_config%d.json" % a.config)))
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(os.path.join(d, "kernel_stats_config%d.csv" % a.config)))}
bench = json.loads(open(os.path.join(d, "bench_config%d_final.json" % a.config)).read().strip().splitlines()[-1])
pairs = bench["config"]["pairs_per_gpu"]
L, k = bench["config"]["read_len"], bench["config"]["k"]
alg = pairs * (2 * ((L + 3) // 4) + 2 * (L - k) * 8 + 16)
steps_pmc = 3  # tools/profile.sh: --steps 2 --warmup 1
rows, total = [], 0.0
for name, v in pmc.items():
    if not isinstance(v, dict) or "FETCH_SIZE" not in v or not name.startswith("k_"):
        continue
    if name in ("k_synth", "k_iota_woff", "k_dense_zero_cnt", "k_seed_fill", "k_seed_insert", "k_table_finalize", "k_pack_nodes"):
        continue  # (index build / block synthesis: not part of the step)
    per = (2.0 * v["FETCH_SIZE"]["per_dispatch_mean"] + v.get("WRITE_SIZE", {}).get("per_dispatch_mean", 0.0)) * 1024.0
    calls_per_step = v["FETCH_SIZE"]["dispatches"] / steps_pmc
    st = stats.get(name)
    ms = float(st["AverageNs"]) / 1e6 if st else None
    rows.append({"kernel": name, "launches_per_step": calls_per_step, "GB_per_launch": per / 1e9, "ms_per_launch": ms,
                 "TB_per_s": (per / 1e12) / (ms / 1e3) if ms else None, "GB_per_step": per * calls_per_step / 1e9})
    total += per * calls_per_step
rows.sort(key=lambda r: -r["GB_per_step"])
out = {"config": a.config, "pairs_per_step": pairs, "algorithmic_GB_per_step": alg / 1e9, "counter_GB_per_step": total / 1e9,
       "counter_bytes_over_algorithmic": total / alg, "ms_per_step": bench["ms_per_step"], "kernels": rows,
       "source": "profiles/r%d/pmc_summary_config%d.json (2 x FETCH_SIZE + WRITE_SIZE) and kernel_stats_config%d.csv" % (a.round, a.config, a.config)}
path = a.out or os.path.join(d, "step_bytes_config%d.json" % a.config)
json.dump(out, open(path, "w"), indent=1)
print("config %d: %.1f GB per step / %.1f GB algorithmic = %.2f x" % (a.config, total / 1e9, alg / 1e9, total / alg))
for r in rows[:14]:
    print("  %-34s x%-4.1f %6.2f GB %7.3f ms %5.2f TB/s" % (r["kernel"][:34], r["launches_per_step"], r["GB_per_launch"], r["ms_per_launch"] or 0, r["TB_per_s"] or 0))
