import os, sys, time, copy, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["VS_GRAPH_INTERPRETED"] = "1"   # so that the module-level lookups can be wrapped
import torch
from vstrains_amd import pe as host, _native as nat
from vstrains_amd.graph import pipeline, hip_ops
from vstrains_amd.graph.hip_ops import HipBackend, HipPeLinks
from vstrains_amd.workloads import CONFIGS, workload
cfg = CONFIGS[2]
wd = tempfile.mkdtemp()
st, pre, names, seqs, cum, logger, _ = workload(wd, k=cfg["k"], n_strains=cfg["n_strains"], genome_len=cfg["genome_len"], snp_rate=cfg["snp_rate"], seed=cfg["seed"], read_len=cfg["read_len"], abundance_ratio=cfg["abundance_ratio"])
ctx = host.Context(0); ctx.build_index(seqs, cfg["k"])
reads = ctx.synth_pairs(st.genomes, cum, 20250002, 0, 2000000, 150, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
counter = host.PeCounter(ctx); counter.add(reads)
lib = nat.lib()
acc = {}
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc[self.name] = acc.get(self.name, 0.0) + time.perf_counter() - t; acc[self.name + "#"] = acc.get(self.name + "#", 0) + 1; return r
class Lib:
    def __getattr__(self, k):
        f = getattr(lib, k)
        return Timed(k, f) if k.startswith("vs_") else f
nat_lib = Lib()
nat.lib = lambda: nat_lib
backend = HipBackend(ctx=ctx)
for rep in range(3):
    acc.clear()
    out = os.path.join(wd, "o%d" % rep)
    for s in ("gfa", "tmp"): os.makedirs(os.path.join(out, s), exist_ok=True)
    p = copy.deepcopy(pre)
    t0 = time.perf_counter()
    table = HipPeLinks.from_counter(ctx, counter, names)
    pipeline.extract_strains(p, table, backend, logger, out)
    tot = time.perf_counter() - t0
    print("total %.3f native %.3f" % (tot, sum(v for k, v in acc.items() if not k.endswith("#"))), {k: (round(v, 4), acc[k + "#"]) for k, v in acc.items() if not k.endswith("#")})
