"""Host side of PE-link inference: the Python mirror of the reference's
``utils/VStrains_PE_Inference.py`` (file:line citations are into /root/reference) on top of the
HIP library.  Torch is used for device buffers, the stream and torch.distributed only."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat


# ---- text inputs ---------------------------------------------------------------------------------
def _universal_newlines(raw: bytes) -> bytes:
    # open(path, "r") translates \r\n and lone \r to \n (PE_Inference.py:105,147-150 use text mode)
    if b"\r" in raw:
        raw = raw.replace(b"\r\n", b"\n").replace(b"\r", b"\n")
    return raw


def read_gfa_segments(path: str) -> Tuple[List[str], List[str]]:
    """``S`` records in file order -> (ids, sequences).  Every line loses its last character
    before the tab split, whatever that character is (PE_Inference.py:105-112)."""
    with open(path, "rb") as fh:
        raw = _universal_newlines(fh.read())
    ids: List[str] = []
    seqs: List[str] = []
    pieces = raw.split(b"\n")
    last_has_nl = raw.endswith(b"\n")
    if last_has_nl:
        pieces = pieces[:-1]
    for i, piece in enumerate(pieces):
        if i == len(pieces) - 1 and not last_has_nl:
            piece = piece[:-1]  # no newline to drop: a real character goes instead
        cols = piece.split(b"\t")
        if cols[0] == b"S":
            ids.append(cols[1].decode("latin-1"))
            seqs.append(cols[2].decode("latin-1"))
    return ids, seqs


class FastqPair:
    """Both FASTQ files mapped and indexed by the library's multi-threaded host ingest
    (``vs_fastq_*``).  ``ctx`` may be None for host-only use (indexing, ``sequence``, ``gather``)."""

    def __init__(self, fwd: str, rve: str, ctx: "Context" = None):
        self._ctx = ctx
        h = C.c_void_p()
        rc = nat.lib().vs_fastq_open(ctx._h if ctx is not None else None, fwd.encode(), rve.encode(), C.byref(h))
        if rc != nat.VS_OK:
            msg = nat.lib().vs_last_error(ctx._h if ctx is not None else None).decode("utf-8", "replace")
            if "cannot open" in msg:
                raise FileNotFoundError(msg)
            if rc == nat.VS_E_UTF8:  # the reference's readlines() raises UnicodeDecodeError, a ValueError (PE_Inference.py:147-152)
                raise ValueError(msg)
            raise nat.NativeError(rc, msg)
        self._h = h
        info = (C.c_uint64 * 3)()
        nat.lib().vs_fastq_info(h, info)
        self.n_pairs = int(info[0])
        self.lines = (int(info[1]), int(info[2]))

    @classmethod
    def open_shard(cls, fwd: str, rve: str, ctx: "Context", rank: int, world: int, all_gather=None) -> "FastqPair":
        """This rank's contiguous block of the pairs without any rank reading a whole file: rank r counts the newlines
        of byte range r of both files (``vs_fastq_count_part``), the counts are exchanged (``all_gather``: a callable
        that takes this rank's list of integers and returns every rank's, default ``torch.distributed``), the total
        ``min(lines_f // 4, lines_r // 4)`` (PE_Inference.py:154) and this rank's record range follow, and only the
        bytes of those records are indexed (``vs_fastq_open_records``).  ``first`` / ``total_pairs`` say where the block
        lies.  gzip files and files with carriage returns are opened whole by every rank (``whole = True``).

        ``all_gather`` is called TWICE per open, on every rank alike and in this order: once with seven integers (the six
        counts and a failure flag) and once with one (the status of the open itself, whole-file or by records).  A caller
        that brings its own must accept lists of either length (it is handed this rank's list and returns the list of
        every rank's lists, rank order)."""
        from .dist import shard_range

        L = nat.lib()
        # A rank that fails (a missing file, a stale count) must not leave its peers waiting in the exchange: the failure
        # travels WITH the counts (a seventh integer) and with a second, one-integer exchange after the open, and every
        # rank raises.
        mine, failure = [], None
        for path in (fwd, rve):
            out = (C.c_uint64 * 3)()
            rc = L.vs_fastq_count_part(path.encode(), rank, world, out)
            if rc != nat.VS_OK and failure is None:
                msg = L.vs_last_error(None).decode("utf-8", "replace")
                failure = FileNotFoundError(msg) if "cannot open" in msg else ValueError(msg) if rc == nat.VS_E_UTF8 else nat.NativeError(rc, msg)
            mine += [int(out[0]), int(out[1]), int(out[2])] if rc == nat.VS_OK else [0, 0, 0]
        mine.append(0 if failure is None else 1)
        if all_gather is None:
            import torch
            import torch.distributed as dist

            def all_gather(vals):
                t = torch.tensor(vals, dtype=torch.int64)
                if dist.get_backend() == "nccl":
                    t = t.cuda()
                got = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(got, t)
                return [[int(x) for x in g.cpu().tolist()] for g in got]

        everyone = all_gather(mine)
        if failure is not None:
            raise failure
        failed = [r for r, vals in enumerate(everyone) if len(vals) > 6 and vals[6]]
        if failed:
            raise RuntimeError("FASTQ open failed on rank(s) %s" % failed)
        flags = 0
        for vals in everyone:
            flags |= vals[2] | vals[5]
        if flags & 3:  # carriage returns or gzip somewhere: every rank opens the files whole and takes its block of them
            # (a rank that fails alone here -- out of memory while inflating, a transient read error -- tells its peers
            # through the same one-integer exchange as below, so that every rank raises instead of one)
            fq, err = None, None
            try:
                fq = cls(fwd, rve, ctx)
            except Exception as e:  # noqa: BLE001 (whatever it is, the peers must hear of it)
                err = e
            status = all_gather([0 if err is None else 1])
            if err is not None:
                raise err
            failed = [r for r, vals in enumerate(status) if vals[0]]
            if failed:
                fq.close()
                raise RuntimeError("FASTQ open failed on rank(s) %s" % failed)
            fq.total_pairs = fq.n_pairs
            fq.first, last = shard_range(fq.n_pairs, rank, world)
            fq.block_offset = fq.first
            fq.n_pairs = last - fq.first
            fq.whole = True
            return fq
        counts = [np.asarray([vals[3 * i] for vals in everyone], dtype=np.uint64) for i in range(2)]
        lines = [int(counts[i].sum()) + (1 if any(vals[3 * i + 2] & 4 for vals in everyone) else 0) for i in range(2)]
        total = min(lines[0] // 4, lines[1] // 4)
        first, last = shard_range(total, rank, world)
        self = cls.__new__(cls)
        self._ctx = ctx
        h = C.c_void_p()
        rc = L.vs_fastq_open_records(ctx._h if ctx is not None else None, fwd.encode(), rve.encode(), world, counts[0].ctypes.data,
                                     counts[1].ctypes.data, first, last, C.byref(h))
        err = None
        if rc != nat.VS_OK:
            msg = L.vs_last_error(ctx._h if ctx is not None else None).decode("utf-8", "replace")
            err = ValueError(msg) if rc == nat.VS_E_UTF8 else nat.NativeError(rc, msg)
        status = all_gather([0 if err is None else 1])
        if err is not None:
            raise err
        failed = [r for r, vals in enumerate(status) if vals[0]]
        if failed:
            if h:
                L.vs_fastq_close(h)
            raise RuntimeError("FASTQ open failed on rank(s) %s" % failed)
        self._h = h
        self.n_pairs = last - first
        self.lines = tuple(lines)
        self.total_pairs, self.first, self.block_offset, self.whole = total, first, 0, False
        return self

    @property
    def bytes_indexed(self) -> int:
        return int(nat.lib().vs_fastq_bytes_indexed(self._h))

    def __len__(self):
        return self.n_pairs

    def close(self):
        if self._h:
            nat.lib().vs_fastq_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _raise(ctx_h, rc):
        msg = nat.lib().vs_last_error(ctx_h).decode("utf-8", "replace")
        if rc == nat.VS_E_UTF8:  # the reference's readlines() raises UnicodeDecodeError, a ValueError (PE_Inference.py:147-152)
            raise ValueError(msg)
        raise nat.NativeError(rc, msg)

    def sequence(self, which: int, record: int) -> str:
        """The CHARACTERS of a sequence line, one byte each ('?' stands for a multi-byte UTF-8 character)."""
        n = C.c_uint32(0)
        rc = nat.lib().vs_fastq_sequence(self._h, which, record, None, 0, C.byref(n))
        if rc == nat.VS_E_UTF8:
            self._raise(None, rc)
        if rc != nat.VS_OK:
            raise IndexError(record)
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        nat.lib().vs_fastq_sequence(self._h, which, record, buf.ctypes.data, buf.size, C.byref(n))
        return bytes(buf[: n.value]).decode("latin-1")

    def gather(self, first: int, count: int) -> Tuple[np.ndarray, np.ndarray]:
        off = np.zeros(2 * count + 1, dtype=np.uint64)
        rc = nat.lib().vs_fastq_gather(self._h, first, count, off.ctypes.data, None)
        if rc != nat.VS_OK:
            self._raise(None, rc)
        data = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        rc = nat.lib().vs_fastq_gather(self._h, first, count, off.ctypes.data, data.ctypes.data)
        if rc != nat.VS_OK:
            self._raise(None, rc)
        return data, off

    def block(self, first: int, count: int) -> "ReadBlock":
        h = C.c_void_p()
        rc = nat.lib().vs_fastq_block(self._ctx._h, self._h, first, count, C.byref(h))
        if rc != nat.VS_OK:
            self._raise(self._ctx._h, rc)  # ValueError for sequence bytes that are not valid UTF-8
        return ReadBlock(self._ctx, h)


def encode_seqs(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    # one byte per CHARACTER (the reference works on decoded text): a character beyond latin-1 becomes '?', which like
    # any byte outside ACGT makes the windows over it miss
    data = np.frombuffer("".join(seqs).encode("latin-1", "replace"), dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(data), off


def node_order(seqs: Sequence[str], ksize: int, _encoded=None) -> np.ndarray:
    """Numbering of the nodes along the graph's paths (``vs_node_order_host``, csrc/vs_order_host.cpp):
    ``order[r]`` = position in ``seqs`` of the node that gets number r."""
    data, off = _encoded if _encoded is not None else encode_seqs(seqs)
    order = np.zeros(max(len(seqs), 1), dtype=np.uint32)
    rc = nat.lib().vs_node_order_host(data.ctypes.data, off.ctypes.data, len(seqs), ksize, order.ctypes.data)
    if rc != nat.VS_OK:
        raise nat.NativeError(rc, "vs_node_order_host")
    return order[: len(seqs)].astype(np.int64)


# ---- device objects ------------------------------------------------------------------------------
class ReadBlock:
    def __init__(self, ctx: "Context", handle):
        self._ctx = ctx
        self._h = handle

    @property
    def info(self):
        a = (C.c_uint64 * 5)()
        nat.check(self._ctx._h, nat.lib().vs_reads_info(self._h, a))
        return dict(ends=a[0], words=a[1], max_len=a[2], invalid_ends=a[3], device_bytes=a[4])

    def unpack(self):
        inf = self.info
        n = inf["ends"]
        lens = np.zeros(max(n, 1), dtype=np.uint32)
        flags = np.zeros(max(n, 1), dtype=np.uint8)
        # two passes: lengths first (cheap), then text
        out = np.zeros(max(int(inf["words"]) * 16, 1), dtype=np.uint8)
        nat.check(self._ctx._h, nat.lib().vs_reads_unpack(self._ctx._h, self._h, out.ctypes.data,
                                                          lens.ctypes.data, flags.ctypes.data))
        total = int(lens[:n].sum())
        return out[:total], lens[:n], flags[:n]

    def free(self):
        if self._h:
            nat.lib().vs_reads_free(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One HIP context (= one device).  ``device`` is the HIP ordinal."""

    def __init__(self, device: int = 0):
        L = nat.lib()
        h = C.c_void_p()
        rc = L.vs_ctx_create(device, C.byref(h))
        if rc != nat.VS_OK:
            nat.check(None, rc)
        self._h = h
        self.device = device
        self.n_nodes = 0
        self.ksize = 0
        self.node_order = None  # internal number -> position in the caller's node list (None: identical)
        self.node_rank = None   # the inverse

    def close(self):
        if self._h:
            nat.lib().vs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        nat.check(self._h, nat.lib().vs_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def sync(self):
        nat.check(self._h, nat.lib().vs_ctx_sync(self._h))

    def build_index(self, seqs: Sequence[str], ksize: int, renumber: Optional[bool] = None):
        """KeyError(char) if a node of length >= ksize+1 holds a byte outside ACGT, as the
        reference's reverse_seq raises (PE_Inference.py:12-13).

        ``renumber`` (default on; ``VS_EXPERIMENT=1 VS_RENUMBER=0`` turns the default off): the index is built over the nodes in an
        order that runs along the graph's paths (``node_order``, csrc/vs_order_host.cpp) instead of the GFA's -- the
        device's matrices are then in that INTERNAL numbering (``node_order[internal] = position in seqs``), and
        everything that hands results out maps back: ``PeCounter.result``, ``map_ends``, ``HipPeLinks``.  Sums do not
        depend on the numbering; the time does (20-25 % of the step on a graph whose numbering scatters neighbours)."""
        if renumber is None:  # (the switch exists in experiment mode only, like the library's: INTEGRATION.md)
            renumber = not (os.environ.get("VS_EXPERIMENT") and os.environ.get("VS_RENUMBER", "1") == "0")
        data, off = encode_seqs(seqs)
        order = None
        if renumber and len(seqs) > 1:
            order = node_order(seqs, ksize, _encoded=(data, off))
            if np.array_equal(order, np.arange(len(seqs), dtype=order.dtype)):
                order = None
        if order is not None:
            data, off = encode_seqs([seqs[i] for i in order.tolist()])
        bad_node = C.c_uint32(0)
        bad_char = C.c_uint8(0)
        rc = nat.lib().vs_index_build(self._h, data.ctypes.data, off.ctypes.data, len(seqs), ksize,
                                      C.byref(bad_node), C.byref(bad_char))
        if rc == nat.VS_E_NODE_BASE:
            if order is not None:  # (the byte the reference names is the first one in GFA order: ask again in that order)
                return self.build_index(seqs, ksize, renumber=False)
            raise KeyError(chr(bad_char.value))
        nat.check(self._h, rc)
        self.n_nodes = len(seqs)
        self.ksize = ksize
        self.node_order = order  # None: the numbering of ``seqs``
        self.node_rank = None
        if order is not None:
            self.node_rank = np.empty(len(seqs), dtype=np.int64)
            self.node_rank[order] = np.arange(len(seqs), dtype=np.int64)

    def internal_cells(self, mat: int, u, v) -> np.ndarray:
        """Flat cell index inside ``PeCounter.mats[mat]`` (internal numbering) of the cells (u, v) given in the
        numbering of ``build_index``'s ``seqs``; ``mat`` 1 = short_mat, whose cell is (smaller, larger) index
        (PE_Inference.py:174-184) in the numbering it was counted in."""
        u = np.asarray(u, dtype=np.int64)
        v = np.asarray(v, dtype=np.int64)
        if self.node_rank is not None:
            u, v = self.node_rank[u], self.node_rank[v]
            if mat == 1:
                u, v = np.minimum(u, v), np.maximum(u, v)
        return u * self.n_nodes + v

    @property
    def index_info(self):
        a = (C.c_uint64 * 6)()
        nat.check(self._h, nat.lib().vs_index_info(self._h, a))
        return dict(seed_len=a[0], stride=a[1], seed_positions=a[2], slots=a[3], distinct_seeds=a[4], device_bytes=a[5])

    def pack(self, ascii_bytes: np.ndarray, off: np.ndarray) -> ReadBlock:
        ascii_bytes = np.ascontiguousarray(ascii_bytes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        nat.check(self._h, nat.lib().vs_reads_pack(self._h, ascii_bytes.ctypes.data, off.ctypes.data,
                                                   off.size - 1, C.byref(h)))
        return ReadBlock(self, h)

    def pack_pairs(self, fwd: Sequence[str], rve: Sequence[str]) -> ReadBlock:
        n = min(len(fwd), len(rve))
        seqs: List[str] = []
        for r in range(n):
            seqs.append(fwd[r])
            seqs.append(rve[r])
        data, off = encode_seqs(seqs)
        return self.pack(data, off)

    def synth_pairs(self, genomes: Sequence[str], cum: np.ndarray, seed: int, first_pair: int, n_pairs: int,
                    read_len: int, sub_thresh: int = 0, n_thresh: int = 0) -> ReadBlock:
        gd, go = encode_seqs(genomes)
        cum = np.ascontiguousarray(cum, dtype=np.uint32)
        h = C.c_void_p()
        nat.check(self._h, nat.lib().vs_synth_pairs(self._h, gd.ctypes.data, go.ctypes.data, cum.ctypes.data,
                                                    len(genomes), seed, first_pair, n_pairs, read_len,
                                                    sub_thresh, n_thresh, C.byref(h)))
        return ReadBlock(self, h)

    def pe_count(self, reads: ReadBlock, node_mat_ptr: int, short_mat_ptr: int, stats_ptr: int, tile_map_ptr: Optional[int] = None):
        if tile_map_ptr is not None:  # also mark the 64 x 64 counter tiles the block adds to
            nat.check(self._h, nat.lib().vs_pe_count_tracked(self._h, reads._h, C.c_void_p(node_mat_ptr), C.c_void_p(short_mat_ptr),
                                                             C.c_void_p(stats_ptr), C.c_void_p(tile_map_ptr)))
            return
        nat.check(self._h, nat.lib().vs_pe_count(self._h, reads._h, C.c_void_p(node_mat_ptr),
                                                 C.c_void_p(short_mat_ptr), C.c_void_p(stats_ptr)))

    def last_timing(self):
        a = (C.c_double * 5)()
        nat.check(self._h, nat.lib().vs_pe_last_timing(self._h, a))
        return dict(main_ms=a[0], slow_ms=a[1], slow_pairs=int(a[2]), sort_ms=a[3], accumulate_ms=a[4])

    @property
    def last_kernel(self) -> str:
        """Name of the mapping-kernel instantiation the last ``pe_count`` launched."""
        return (nat.lib().vs_pe_last_kernel(self._h) or b"").decode()

    RAN_LOCUS_LDS_SORT, RAN_LOCUS_GLOBAL_SORT, RAN_PE_MID, RAN_ROW_OWNERS = 1, 2, 8, 16

    @property
    def last_launched(self) -> int:
        """VS_RAN_* bits: which optional kernels the most recent ``pe_count`` launched."""
        return int(nat.lib().vs_pe_last_launched(self._h))

    def map_ends(self, reads: ReadBlock, cap: int = 64) -> List[List[int]]:
        n = reads.info["ends"]
        lists = np.zeros((max(n, 1), cap), dtype=np.uint32)
        counts = np.zeros(max(n, 1), dtype=np.uint32)
        nat.check(self._h, nat.lib().vs_pe_map_ends(self._h, reads._h, cap, lists.ctypes.data, counts.ctypes.data))
        out = []
        for e in range(n):
            c = int(counts[e])
            if c > cap:
                raise ValueError("end %d has %d nodes, cap %d" % (e, c, cap))
            ids = lists[e, :c] if self.node_order is None else self.node_order[lists[e, :c]]
            out.append(sorted(int(x) for x in ids))
        return out


U32_LIMIT = 2 ** 32


class PeCounter:
    """node_mat / short_mat accumulation on one device (PE_Inference.py:137-188), counters held
    in torch tensors so that torch.distributed (RCCL) can all-reduce them in place.

    The kernels count in uint32 cells (``mats``: int32 storage, the bits are what matters -- a
    two's-complement sum IS the uint32 sum).  A cell grows by at most 2 per pair (``short_mat[i][i]``
    takes one increment per end, :174-184), so a buffer is exact while ``2 * pairs < 2**32``; before
    that bound is reached the buffer is folded into int64 totals on the device (``vs_counts_fold``),
    the reference's own cell type (``numpy.zeros(..., dtype=int)``, :139-140)."""

    # counters at least this large keep a map of the 64 x 64 tiles their blocks touch, and ``reset`` zeroes those tiles
    # only (vs_pe_count_tracked / vs_counts_zero_tracked); below it clearing the whole buffer is the cheaper way
    TRACK_TILES_MIN_BYTES = 2 << 30

    def __init__(self, ctx: Context, device: Optional[str] = None, track_tiles: Optional[bool] = None):
        import torch

        self.torch = torch
        self.ctx = ctx
        self.device = torch.device(device or ("cuda:%d" % ctx.device))
        n = ctx.n_nodes
        self.n = n
        self.mats = torch.zeros((2, max(n, 1), max(n, 1)), dtype=torch.int32, device=self.device)
        if track_tiles is None:  # (VS_TRACK_TILES=1: tests run the tracked path on small graphs)
            track_tiles = self.device.type == "cuda" and (self.mats.numel() * 4 >= self.TRACK_TILES_MIN_BYTES or os.environ.get("VS_TRACK_TILES") == "1")
        tiles = (max(n, 1) + 63) // 64
        self.tile_map = torch.zeros(2 * tiles * tiles, dtype=torch.uint8, device=self.device) if track_tiles else None
        self.stats = torch.zeros(3, dtype=torch.int64, device=self.device)
        self.wide = None          # int64 totals, allocated by the first fold
        self.pairs_in_buffer = 0  # pairs counted into ``mats`` since it was last empty (all ranks, after a sum)
        self.pairs_seen = 0
        self.last_all_reduce = None  # "dense" / "compact" after all_reduce()
        # (the numbering of the index this counter counts under: kept here, so that a later build_index on the same context
        # cannot change how these matrices are read)
        self.node_order = getattr(ctx, "node_order", None)
        self.node_rank = getattr(ctx, "node_rank", None)

    def reserve_link_table(self) -> None:
        """Set aside the device buffer of the PE-link table the graph stages will build from these counters (N x N int64,
        ``vs_links_reserve``): called where the counters are made when the stages follow, so that the table's build does
        not wait for a large ``hipMalloc`` (0.3 ms or half a second for 23.7 GB, depending on what was freed before).
        An optimisation only: when the memory is not there now, the build allocates later (or fails there, where every rank
        can be told) -- a lone rank raising here would leave its peers waiting in the next collective."""
        if self.tile_map is not None and self.n >= 32768:
            return  # (the table of such counters is built from their dirty tiles as CSR rows, ABI 10: nothing to set aside)
        rc = nat.lib().vs_links_reserve(self.ctx._h, self.n)
        if rc == nat.VS_E_OOM:
            import warnings

            warnings.warn("the PE-link table's buffer could not be set aside now (%d nodes); it is allocated when the table is built" % self.n)
            return
        nat.check(self.ctx._h, rc)

    def reset(self):
        if self.tile_map is not None:
            # only the tiles the blocks since the last reset touched (every add marks them, every sum over the ranks ORs the
            # ranks' maps): the invariant is that a cell outside the marked tiles is zero
            torch = self.torch
            with torch.cuda.device(self.device):
                self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                nat.check(self.ctx._h, nat.lib().vs_counts_zero_tracked(self.ctx._h, C.c_void_p(self.mats[0].data_ptr()),
                                                                        C.c_void_p(self.mats[1].data_ptr()), self.n,
                                                                        C.c_void_p(self.tile_map.data_ptr())))
        else:
            self.mats.zero_()
        self.stats.zero_()
        if self.wide is not None:
            self.wide.zero_()
        self.pairs_in_buffer = 0
        self.pairs_seen = 0

    def fold(self):
        """uint32 buffer -> int64 totals (device), buffer left empty."""
        torch = self.torch
        if self.wide is None:
            self.wide = torch.zeros(self.mats.shape, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            nat.check(self.ctx._h, nat.lib().vs_counts_fold(self.ctx._h, C.c_void_p(self.mats.data_ptr()),
                                                            C.c_void_p(self.wide.data_ptr()), self.mats.numel()))
        if self.tile_map is not None:
            self.tile_map.zero_()  # (the fold left every uint32 cell at zero)
        self.pairs_in_buffer = 0

    def add(self, reads: ReadBlock):
        torch = self.torch
        n_pairs = reads.info["ends"] // 2
        if 2 * (self.pairs_in_buffer + n_pairs) >= U32_LIMIT:
            self.fold()
        if 2 * n_pairs >= U32_LIMIT:
            raise OverflowError("one read block of %d pairs can overflow a uint32 cell; use blocks of < 2^31 pairs" % n_pairs)
        self.pairs_in_buffer += n_pairs
        self.pairs_seen += n_pairs
        with torch.cuda.device(self.device):
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            self.ctx.pe_count(reads, self.mats[0].data_ptr(), self.mats[1].data_ptr(), self.stats.data_ptr(),
                              tile_map_ptr=self.tile_map.data_ptr() if self.tile_map is not None else None)

    def _occupied(self, head):
        """uint8 [m]: which 64-cell stretches of ``head`` ([m, 64] view of a counter buffer on this device) hold a non-zero
        cell -- one pass of the library's kernel over the buffer, on torch's current stream."""
        torch = self.torch
        occ = torch.empty(head.shape[0], dtype=torch.uint8, device=head.device)
        with torch.cuda.device(self.device):
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            nat.check(self.ctx._h, nat.lib().vs_counts_occupied(self.ctx._h, C.c_void_p(head.data_ptr()), head.element_size(),
                                                                head.shape[0], C.c_void_p(occ.data_ptr())))
        return occ

    def all_reduce(self, dst=None, predict=False):
        """Sum over the ranks of the process group, in place, in TWO collectives (``dist.sum_counts_packed``): a MAX over
        [occupancy of the counters' 64-cell stretches | flag bytes], then a SUM over [the occupied stretches of the union |
        a row with the three stats and the pairs in the buffers].  The flag bytes settle, for every rank alike, whether the
        uint32 buffers can hold the sum (a bound on the largest rank's pairs, times the ranks): if not, or if some rank
        already holds int64 totals, every rank folds and the int64 totals are summed (a rare second exchange); a rank
        whose environment turns the compact exchange off vetoes it for everybody (no mismatched collectives).

        ``dst``: only that rank needs the sums (the drop-in's writer): the SUM collectives reduce to it, the other ranks
        keep their own counts.  ``predict``: the steady state of fixed-size steps (bench.py) -- no host wait at all; the
        preconditions (no int64 totals, the bound, the veto) are checked statically here and again from the flag bytes when
        ``settle()`` is called before the buffer is reused."""
        from .dist import group_size

        if group_size() <= 1:
            return
        from . import dist as vdist

        torch = self.torch
        world = group_size()
        may_compact = os.environ.get("VS_COMPACT_ALLREDUCE", "1") not in ("0", "")
        p = int(self.pairs_in_buffer)
        a_bits = p.bit_length()
        flags = torch.zeros(vdist.FLAG_BYTES, dtype=torch.uint8)
        flags[vdist.FLAG_NO_COMPACT] = 0 if may_compact else 1
        flags[vdist.FLAG_WIDE] = 1 if self.wide is not None else 0
        # the largest rank's pairs as (bit length, its top eight bits): MAX over the bytes bounds the maximum from above
        flags[vdist.FLAG_PAIRS_LOG2] = a_bits
        flags[vdist.FLAG_PAIRS_LOG2 + 1] = (p >> (a_bits - 8)) if a_bits > 8 else p
        tail = torch.cat([self.stats.to(torch.int64), torch.tensor([p], dtype=torch.int64, device=self.stats.device)])
        if not hasattr(self, "_xstate"):
            self._xstate = vdist.ExchangeState()
        timing = {} if os.environ.get("VS_DIST_TIMING") else None
        occ_fn = self._occupied if self.mats.is_cuda else None
        if predict and self._xstate.cap is not None:
            if self.wide is not None or 2 * p * world >= U32_LIMIT or not may_compact:
                raise OverflowError("a predicted exchange sums uint32 buffers through their occupied stretches: 2 * %d pairs * %d "
                                    "ranks must fit, without int64 totals or a veto; use all_reduce()" % (p, world))
            how, _, out_tail = vdist.sum_counts_packed(self.mats, tail, flags, tile_map=self.tile_map, timing=timing, occupancy_fn=occ_fn,
                                                       state=self._xstate, predict=True, dst=dst)
            self.last_all_reduce = how
            self.stats.copy_(out_tail[:3])
            self.pairs_in_buffer = p * world  # (a bound; the steps' blocks are of one size)
        else:
            how, fl, out_tail = vdist.sum_counts_packed(self.mats, tail, flags, tile_map=self.tile_map, timing=timing, occupancy_fn=occ_fn,
                                                        state=self._xstate, dst=dst, on_flags=self._fold_if_needed)
            self.last_all_reduce = how
            out = out_tail.cpu()
            self.stats.copy_(out[:3].to(self.stats.device))
            self.pairs_in_buffer = 0 if self.wide is not None else int(out[3])
        self.last_collectives = self._xstate.collectives
        if timing is not None:
            self.exchange_timing = getattr(self, "exchange_timing", [])
            self.exchange_timing.append(timing)

    def _fold_if_needed(self, fl):
        """The flag bytes of all ranks are in: do the uint32 buffers hold the sum?  If not, every rank folds and the int64
        totals are what is summed (returned to ``sum_counts_packed``, which starts over on them)."""
        from . import dist as vdist

        a_bits, top = int(fl[vdist.FLAG_PAIRS_LOG2]), int(fl[vdist.FLAG_PAIRS_LOG2 + 1])
        bound = ((top + 1) << (a_bits - 8)) if a_bits > 8 else 255
        if fl[vdist.FLAG_WIDE] or 2 * bound * vdist.group_size() >= U32_LIMIT:
            self.fold()
            return self.wide
        return None

    def _xstate_cap(self, value=None):
        """The union size the next predicted exchange is staged for (None: not known yet); with a value: set it."""
        from . import dist as vdist

        if not hasattr(self, "_xstate"):
            self._xstate = vdist.ExchangeState()
        if value is not None:
            self._xstate.cap = value
        return self._xstate.cap

    def settle(self):
        """The deferred half of ``all_reduce(predict=True)``; call before the buffer is counted into again."""
        from . import dist as vdist

        if getattr(self, "_xstate", None) is not None:
            vdist.settle_exchange(self._xstate)

    def all_reduce_async(self):
        """Overlapped form for fixed-size steps (bench.py): the caller keeps counting into a second
        buffer meanwhile.  No agreement round, so the bound is checked statically."""
        from .dist import all_reduce_counts_async, group_size

        world = max(group_size(), 1)
        if self.wide is not None or 2 * self.pairs_in_buffer * world >= U32_LIMIT:
            raise OverflowError("all_reduce_async sums uint32 buffers: 2 * %d pairs * %d ranks does not fit; use all_reduce()"
                                % (self.pairs_in_buffer, world))
        work = all_reduce_counts_async(self.mats, self.stats)
        if self.tile_map is not None:
            import torch.distributed as dist

            work.append(dist.all_reduce(self.tile_map, op=dist.ReduceOp.MAX, async_op=True))
        self.pairs_in_buffer *= world
        return work

    def user_order(self, t):
        """[2,N,N] device tensor in the index's internal numbering -> the caller's (``Context.build_index``):
        node_mat rows and columns permuted; short_mat, which holds a pair of nodes at (smaller, larger) number
        (PE_Inference.py:174-184: ``i <= i2`` over ascending indices), mirrored first and cut back to the upper
        triangle of the caller's numbering after."""
        torch = self.torch
        rank = self.node_rank
        if rank is None:
            return t
        r = torch.from_numpy(rank).to(t.device)
        # (one matrix at a time, every intermediate freed before the next is made: a 54 k-node graph is 11.8 GB per
        # int32 matrix, and this runs when a run that counted fine wants its result)
        out = torch.empty_like(t)
        rows = t[0].index_select(0, r)
        torch.index_select(rows, 1, r, out=out[0])
        del rows
        s = t[1] + t[1].t()
        s.diagonal().sub_(t[1].diagonal())  # (the diagonal was doubled by the mirror)
        rows = s.index_select(0, r)
        del s
        torch.index_select(rows, 1, r, out=out[1])
        del rows
        out[1].triu_()
        return out

    def result(self):
        """-> (node_mat int64 [N,N], short_mat int64 [N,N], (n_reads, short_reads, used_reads)), in the numbering of
        the node list ``Context.build_index`` was given."""
        n = self.n
        m = self.user_order(self.mats[:, :n, :n]).cpu().numpy().view(np.uint32).astype(np.int64)
        if self.wide is not None:
            m += self.user_order(self.wide[:, :n, :n]).cpu().numpy()
        s = self.stats.cpu().numpy()
        return m[0], m[1], (int(s[0]), int(s[1]), int(s[2]))


# ---- outputs -------------------------------------------------------------------------------------
def write_matrix_text(path: str, ids: Sequence[str], mat: np.ndarray):
    """``{id_i}:{id_j}:{count}`` for all i, j in row-major order, zeros included
    (PE_Inference.py:194-205) -- N^2 lines, formatted by the library on all host cores."""
    names = [s.encode("latin-1") for s in ids]
    off = np.zeros(len(names) + 1, dtype=np.uint64)
    if names:
        off[1:] = np.cumsum([len(b) for b in names], dtype=np.uint64)
    blob = np.frombuffer(b"".join(names) or b"\0", dtype=np.uint8)
    m = np.ascontiguousarray(mat, dtype=np.int64)
    rc = nat.lib().vs_write_matrix_text(None, path.encode(), blob.ctypes.data, off.ctypes.data, len(names),
                                        m.ctypes.data if m.size else None)
    if rc != nat.VS_OK:
        raise OSError(nat.lib().vs_last_error(None).decode("utf-8", "replace"))
