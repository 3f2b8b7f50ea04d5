#!/usr/bin/env python3
"""GPU-side statistics for the counter kernel: after ONE step of a config, how many distinct counter cells the block
touched (the floor for any scheme that sums increments before they reach memory), how they spread over 64-byte stretches,
64 x 64 tiles and strips of rows (index numbering, the one the kernels see), and how wide the band is.
    python tools/cell_probe.py [config] [pairs]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vstrains_amd import pe as host  # noqa: E402
from vstrains_amd.workloads import CONFIGS, workload_for  # noqa: E402


def q(t, qs=(0.5, 0.9, 0.99, 1.0)):
    t = t.to(torch.float64).flatten().sort().values
    return [int(t[min(len(t) - 1, int(p * (len(t) - 1)))].item()) for p in qs]


def main():
    cfg_i = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cfg = CONFIGS[cfg_i]
    M = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["total_pairs"] // cfg["gpus"]
    st, pre, names, seqs, cum, logger, _ = workload_for(cfg_i, tempfile.mkdtemp())
    ctx = host.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.build_index(seqs, cfg["k"])
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + cfg_i, 0, M, cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    c = host.PeCounter(ctx)
    c.add(reads)
    torch.cuda.synchronize()
    n = c.n
    print("config %d: %d nodes, %d pairs" % (cfg_i, n, M))
    for mi, name in ((0, "node_mat"), (1, "short_mat")):
        m = c.mats[mi]
        total = int(m.to(torch.int64).sum().item())
        nzrow = torch.zeros(n, dtype=torch.int64, device=m.device)
        lo = torch.full((n,), n, dtype=torch.int64, device=m.device)
        hi = torch.zeros(n, dtype=torch.int64, device=m.device)
        stretches = 0
        cols = torch.arange(n, device=m.device)
        slab = 2048
        for r0 in range(0, n, slab):
            z = m[r0:r0 + slab] != 0
            nzrow[r0:r0 + slab] = z.sum(dim=1)
            anyr = z.any(dim=1)
            first = torch.where(z, cols[None, :], n).min(dim=1).values
            last = torch.where(z, cols[None, :], -1).max(dim=1).values
            lo[r0:r0 + slab] = first
            hi[r0:r0 + slab] = torch.where(anyr, last, torch.zeros_like(last))
            w = (n // 16) * 16
            stretches += int(z[:, :w].view(z.shape[0], -1, 16).any(dim=2).sum().item())
        D = int(nzrow.sum().item())
        print("%s: increments %d, distinct cells %d (%.1f increments per cell), non-zero 64-byte stretches %d (%.2f cells each)" % (
            name, total, D, total / max(D, 1), stretches, D / max(stretches, 1)))
        print("   cells per row: 50/90/99/max %s; rows with cells %d" % (q(nzrow), int((nzrow > 0).sum().item())))
        span = torch.where(nzrow > 0, hi - lo + 1, torch.zeros_like(hi))
        rows = torch.arange(n, device=m.device)
        off = torch.where(nzrow > 0, torch.maximum((rows - lo).abs(), (hi - rows).abs()), torch.zeros_like(hi))
        print("   column span of a row 50/90/99/max %s; farthest column from the diagonal 50/90/99/max %s" % (q(span), q(off)))
        for rows_per in (8, 16, 32, 64):
            k = (n + rows_per - 1) // rows_per
            pad = torch.zeros(k * rows_per, dtype=torch.int64, device=m.device)
            pad[:n] = nzrow
            s = pad.view(k, rows_per).sum(dim=1)
            print("   strips of %2d rows: %d strips, cells per strip 50/90/99/max %s" % (rows_per, k, q(s)))
        T = (n + 63) // 64
        tiles = torch.zeros(T, T, dtype=torch.int64, device=m.device)
        for r0 in range(0, n, 64):
            z = (m[r0:r0 + 64] != 0).sum(dim=0)
            padc = torch.zeros(T * 64, dtype=torch.int64, device=m.device)
            padc[:n] = z
            tiles[r0 // 64] = padc.view(T, 64).sum(dim=1)
        nt = int((tiles > 0).sum().item())
        print("   touched 64 x 64 tiles %d of %d, cells per touched tile 50/90/99/max %s" % (nt, T * T, q(tiles[tiles > 0])))
        # tiles per tile row, and how far from the diagonal tile
        per_row = (tiles > 0).sum(dim=1)
        print("   touched tiles per tile row 50/90/99/max %s" % q(per_row))


if __name__ == "__main__":
    main()
