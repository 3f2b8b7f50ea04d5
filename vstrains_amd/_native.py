"""ctypes binding of libvstrains_hip.so (include/vstrains_hip.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvstrains_hip.so")

VS_OK = 0
VS_E_ARG, VS_E_HIP, VS_E_OOM, VS_E_NODE_BASE, VS_E_STATE, VS_E_RANGE, VS_E_UTF8, VS_E_KEY, VS_E_FPE = -1, -2, -3, -4, -5, -6, -7, -8, -9

# name -> (restype, argtypes); every symbol include/vstrains_hip.h declares
SYMBOLS = {
    "vs_abi_version": (C.c_int, []),
    "vs_device_count": (C.c_int, []),
    "vs_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vs_ctx_destroy": (None, [C.c_void_p]),
    "vs_last_error": (C.c_char_p, [C.c_void_p]),
    "vs_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_ctx_sync": (C.c_int, [C.c_void_p]),
    "vs_index_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]),
    "vs_index_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "vs_node_order_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "vs_reads_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "vs_reads_free": (None, [C.c_void_p, C.c_void_p]),
    "vs_reads_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "vs_reads_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vs_pack_sequence": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "vs_fastq_open": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "vs_fastq_close": (None, [C.c_void_p]),
    "vs_fastq_count_part": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "vs_fastq_open_records": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_void_p)]),
    "vs_fastq_bytes_indexed": (C.c_uint64, [C.c_void_p]),
    "vs_fastq_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "vs_fastq_sequence": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "vs_fastq_gather": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "vs_fastq_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "vs_write_matrix_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vs_synth_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                 C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_void_p)]),
    "vs_pe_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vs_pe_count_tracked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vs_counts_zero_tracked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vs_counts_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "vs_counts_occupied": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]),
    "vs_comm_unique_id": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vs_comm_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_pe_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "vs_pe_map_ends": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vs_pe_last_kernel": (C.c_char_p, [C.c_void_p]),
    "vs_pe_last_launched": (C.c_uint32, [C.c_void_p]),
    "vs_pe_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "vs_links_from_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "vs_links_from_counts_tracked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "vs_links_from_wide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "vs_links_from_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "vs_links_reserve": (C.c_int, [C.c_void_p, C.c_uint32]),
    "vs_links_free": (None, [C.c_void_p, C.c_void_p]),
    "vs_links_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "vs_links_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vs_links_block_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_uint64, C.c_void_p]),
    "vs_links_group_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vs_graph_refresh": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 13 + [C.POINTER(C.c_uint32)]),
    "vs_stage_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "vs_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vs_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vs_dev_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vs_dev_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
}

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libvstrains_hip: %s (code %d)" % (msg, code))
        self.code = code


def lib():
    """Load the shared object (once).  Raises if it is absent: build it with
    ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C vstrains_amd/csrc``."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: the HIP library is not built and there is no fallback path" % LIB_PATH)
        # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both own
        # the GPU.  Importing torch first makes the loader bind our NEEDED libamdhip64.so.7 to
        # the copy that is already mapped, so the library and torch share one runtime.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        from .graph import native_stage

        native_stage.bind(L)  # the vs_stage_* entry points (their prototypes live next to the blob layout)
        _lib = L
    return _lib


def check(ctx, rc: int):
    if rc != VS_OK:
        msg = lib().vs_last_error(ctx)
        raise NativeError(rc, msg.decode("utf-8", "replace") if msg else "?")
