"""Experiment: does the numbering of the nodes matter to the counter kernel?  Counts the same synthetic block of
configs[CONFIG] with the GFA's numbering, with vstrains_amd.node_order.locality_order and with a random one; prints the
library's per-kernel times and checks that the matrices are the same up to the permutation.

    python tools/order_probe.py [config] [pairs]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vstrains_amd import pe as host  # noqa: E402
from vstrains_amd.node_order import locality_order  # noqa: E402
from vstrains_amd.workloads import CONFIGS, workload_for  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = CONFIGS[config]
R = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["total_pairs"] // cfg["gpus"]
st, pre, names, seqs, cum, logger, _ = workload_for(config, tempfile.mkdtemp(prefix="order_probe_"))
k, L = cfg["k"], cfg["read_len"]
n = len(seqs)
t0 = time.time()
loc = locality_order(seqs, k)
print("locality_order: %.2f s for %d nodes" % (time.time() - t0, n), flush=True)
rng = np.random.default_rng(5)
from vstrains_amd.node_order import path_order
orders = {"gfa": list(range(n)), "locality": loc, "path": path_order(seqs, k), "native path": host.node_order(seqs, k).tolist(), "random": [int(x) for x in rng.permutation(n)]}
if os.environ.get("ORDER_FROM_RANDOM") == "1":  # the orders computed from a randomly numbered graph, as an assembler would hand it over
    rp = orders["random"]
    sq = [seqs[i] for i in rp]
    orders["locality(random)"] = [rp[j] for j in locality_order(sq, k)]
    orders["path(random)"] = [rp[j] for j in path_order(sq, k)]
seed = 20250000 + config
sums = {}
ref = None
for tag, order in orders.items():
    ctx = host.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.build_index([seqs[i] for i in order], k, renumber=False)
    reads = ctx.synth_pairs(st.genomes, cum, seed, 0, R, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    c = host.PeCounter(ctx)
    rows = []
    for it in range(4):
        c.reset()
        c.add(reads)
        t = ctx.last_timing()
        rows.append(t)
    t = rows[-1]
    torch.cuda.synchronize()
    print("%-9s main %.2f  counters %.2f  sort %.2f  overflow %.2f  (%s)" % (
        tag, t["main_ms"], t["accumulate_ms"], t["sort_ms"], t["slow_ms"], ctx.last_kernel), flush=True)
    tot = (int(c.mats[0].sum(dtype=torch.int64)), int(c.mats[1].sum(dtype=torch.int64)), tuple(c.stats.cpu().tolist()))
    # the permuted matrices, sampled: 2 000 random rows against the gfa numbering
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.tensor(order)] = torch.arange(n)
    pick = torch.tensor(np.random.default_rng(9).choice(n, size=min(n, 2000), replace=False))
    sample = []
    for m in (0, 1):
        sub = c.mats[m][rank[pick].cuda()][:, rank.cuda()].cpu()
        if m == 1:  # short_mat holds (min id, max id): symmetrise before comparing numberings
            subT = c.mats[m][:, rank[pick].cuda()].t()[:, rank.cuda()].cpu()
            diag = torch.zeros_like(sub)
            diag[torch.arange(len(pick)), pick] = sub[torch.arange(len(pick)), pick]
            sub = sub + subT - diag
        sample.append(sub)
    if ref is None:
        ref = (tot, sample)
    else:
        ok = tot == ref[0] and all(torch.equal(a, b) for a, b in zip(sample, ref[1]))
        print("   same counts as the gfa numbering: %s" % ok, flush=True)
    del c, reads, ctx
    torch.cuda.empty_cache()
