# cython: language_level=3, boundscheck=False, wraparound=False, initializedcheck=False
"""C-level front half of ``HipGraphOps.reinit`` (``hip_ops.py``): which vertices and edges of a stage
survive (map order, by name -- ``graph_to_gfa``, IO.py:345-369), the arrays ``vs_stage_rebuild`` wants,
and the bytes of the stage GFA, in two passes over the graph's Python lists without a Python-level
loop.  Same results as the Python statement of it in ``hip_ops.py`` (which stays the fallback when
this module is not built; ``tests/test_graph_gpu.py`` compares the two): per stage of 5 000 nodes the
Python loops cost 1.5 ms, 116 stages per run at configs[2], thousands at configs[3] / [4].

A segment line is ``S <id> <seq> DP:f:<repr(depth)>``, a link line ``L <a> + <b> + <overlap>M``.  The
depth's repr comes out of ``dp_repr`` (float -> bytes, filled on first sight; the caller keeps it
across stages), everything else is copied straight from the strings' UTF-8 buffers."""
from cpython.bytes cimport PyBytes_AS_STRING, PyBytes_FromStringAndSize, PyBytes_GET_SIZE
from cpython.dict cimport PyDict_GetItem, PyDict_Next, PyDict_SetItem
from cpython.list cimport PyList_GET_ITEM, PyList_GET_SIZE
from cpython.long cimport PyLong_AsLong, PyLong_AsSsize_t
from cpython.object cimport PyObject_IsTrue
from cpython.ref cimport PyObject
from cpython.tuple cimport PyTuple_GET_ITEM
from libc.stdlib cimport free, malloc
from libc.string cimport memcpy

import numpy as np

cdef extern from "Python.h":
    const char *PyUnicode_AsUTF8AndSize(object unicode, Py_ssize_t *size) except NULL

cdef extern from "stdio.h":
    int snprintf(char *buf, size_t n, const char *fmt, ...)


cdef struct Piece:
    const char *p
    Py_ssize_t n


def prepare(list vblack, list vid, list vdp, list vseq, list eblack, list eovl, list esrc, list etgt, dict nodes, dict edges,
            dict dp_repr):
    """-> (n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text)"""
    cdef Py_ssize_t pos = 0, v, e, i, nv = 0, n_e = 0, ln, total = 0
    cdef PyObject *k
    cdef PyObject *val
    cdef PyObject *hit
    cdef PyObject *hs
    cdef PyObject *ht
    cdef Py_ssize_t n_nodes = len(nodes), n_edges = len(edges)
    cdef list n_vid = [], n_vdp = [], n_vseq = [], kept_keys = [], src = [], tgt = [], ovl = []
    cdef dict nn = {}
    cdef list alive = []   # reprs that are not in the cache, until the text is built
    cdef object name, seq, dp, rep, key, a, b, o
    a_src = np.empty(max(n_edges, 1), dtype=np.uint32)
    a_tgt = np.empty(max(n_edges, 1), dtype=np.uint32)
    a_dp = np.empty(max(n_nodes, 1), dtype=np.float64)
    cdef unsigned int[::1] srcv = a_src
    cdef unsigned int[::1] tgtv = a_tgt
    cdef double[::1] dpv = a_dp
    cdef Piece *seg = <Piece *>malloc(sizeof(Piece) * 3 * (n_nodes + 1))
    cdef Piece *lnk = <Piece *>malloc(sizeof(Piece) * 2 * (n_edges + 1))
    cdef long *ovl_c = <long *>malloc(sizeof(long) * (n_edges + 1))
    cdef char numbuf[32]
    cdef char *out
    cdef int nd
    # An edge's ends are looked up by NAME (the reference goes through the GFA text).  With tens of
    # thousands of nodes those two dict lookups per edge are cache misses and most of the pass, so the
    # common case goes by index: the edge's source vertex carries the very string the key holds, that
    # vertex was kept, and no two kept vertices share an id -- then nn[name] is that vertex's new index.
    cdef Py_ssize_t n_old = PyList_GET_SIZE(vid), sv, tv, si, ti
    cdef int *new_of_old = <int *>malloc(sizeof(int) * (n_old + 1))
    cdef bint by_index
    if seg == NULL or lnk == NULL or ovl_c == NULL or new_of_old == NULL:
        free(seg); free(lnk); free(ovl_c); free(new_of_old)
        raise MemoryError()
    for i in range(n_old):
        new_of_old[i] = -1
    try:
        # ---- vertices: map order, black ones
        while PyDict_Next(nodes, &pos, &k, &val):
            v = PyLong_AsSsize_t(<object>val)
            if not PyObject_IsTrue(<object>PyList_GET_ITEM(vblack, v)):
                continue
            name = <object>PyList_GET_ITEM(vid, v)
            seq = <object>PyList_GET_ITEM(vseq, v)
            dp = <object>PyList_GET_ITEM(vdp, v)
            if type(name) is not str or type(seq) is not str:
                raise TypeError("vertex id / sequence is not a str")  # (the caller takes the Python path)
            n_vid.append(name)
            n_vdp.append(dp)
            n_vseq.append(seq)
            dpv[nv] = dp
            PyDict_SetItem(nn, name, nv)
            new_of_old[v] = <int>nv
            if dpv[nv] == 0.0:   # 0.0 and -0.0 are one dict key and two reprs: not through the cache
                rep = repr(dp).encode()
                alive.append(rep)
            else:
                hit = PyDict_GetItem(dp_repr, dp)
                if hit == NULL:
                    rep = repr(dp).encode()
                    dp_repr[dp] = rep
                else:
                    rep = <object>hit
            seg[3 * nv].p = PyUnicode_AsUTF8AndSize(name, &ln)
            seg[3 * nv].n = ln
            seg[3 * nv + 1].p = PyUnicode_AsUTF8AndSize(seq, &ln)
            seg[3 * nv + 1].n = ln
            seg[3 * nv + 2].p = PyBytes_AS_STRING(rep)   # (alive: dp_repr holds it)
            seg[3 * nv + 2].n = PyBytes_GET_SIZE(rep)
            total += 2 + seg[3 * nv].n + 1 + seg[3 * nv + 1].n + 6 + seg[3 * nv + 2].n + 1
            nv += 1
        # ---- edges: map order, both ends among the kept names, black
        by_index = len(nn) == nv   # (no id twice among the kept vertices)
        pos = 0
        while PyDict_Next(edges, &pos, &k, &val):
            key = <object>k
            e = PyLong_AsSsize_t(<object>val)
            if not PyObject_IsTrue(<object>PyList_GET_ITEM(eblack, e)):
                continue
            a = <object>PyTuple_GET_ITEM(key, 0)
            b = <object>PyTuple_GET_ITEM(key, 1)
            si = -1
            ti = -1
            if by_index:
                sv = PyLong_AsSsize_t(<object>PyList_GET_ITEM(esrc, e))
                tv = PyLong_AsSsize_t(<object>PyList_GET_ITEM(etgt, e))
                if 0 <= sv < n_old and PyList_GET_ITEM(vid, sv) == PyTuple_GET_ITEM(key, 0):
                    si = new_of_old[sv]
                if 0 <= tv < n_old and PyList_GET_ITEM(vid, tv) == PyTuple_GET_ITEM(key, 1):
                    ti = new_of_old[tv]
            if si < 0:
                hs = PyDict_GetItem(nn, a)
                if hs == NULL:
                    continue
                si = PyLong_AsSsize_t(<object>hs)
            if ti < 0:
                ht = PyDict_GetItem(nn, b)
                if ht == NULL:
                    continue
                ti = PyLong_AsSsize_t(<object>ht)
            o = <object>PyList_GET_ITEM(eovl, e)
            src.append(si)
            tgt.append(ti)
            ovl.append(o)
            kept_keys.append(key)
            srcv[n_e] = <unsigned int>si
            tgtv[n_e] = <unsigned int>ti
            ovl_c[n_e] = PyLong_AsLong(o)
            if type(a) is not str or type(b) is not str:
                raise TypeError("edge key holds a non-string id")
            lnk[2 * n_e].p = PyUnicode_AsUTF8AndSize(a, &ln)
            lnk[2 * n_e].n = ln
            lnk[2 * n_e + 1].p = PyUnicode_AsUTF8AndSize(b, &ln)
            lnk[2 * n_e + 1].n = ln
            total += 2 + lnk[2 * n_e].n + 3 + lnk[2 * n_e + 1].n + 3 + snprintf(numbuf, 32, "%ld", ovl_c[n_e]) + 2
            n_e += 1
        # ---- the text
        text = PyBytes_FromStringAndSize(NULL, total)
        out = PyBytes_AS_STRING(text)
        for i in range(nv):
            out[0] = b'S'; out[1] = b'\t'; out += 2
            memcpy(out, seg[3 * i].p, seg[3 * i].n); out += seg[3 * i].n
            out[0] = b'\t'; out += 1
            memcpy(out, seg[3 * i + 1].p, seg[3 * i + 1].n); out += seg[3 * i + 1].n
            memcpy(out, b"\tDP:f:", 6); out += 6
            memcpy(out, seg[3 * i + 2].p, seg[3 * i + 2].n); out += seg[3 * i + 2].n
            out[0] = b'\n'; out += 1
        for i in range(n_e):
            out[0] = b'L'; out[1] = b'\t'; out += 2
            memcpy(out, lnk[2 * i].p, lnk[2 * i].n); out += lnk[2 * i].n
            memcpy(out, b"\t+\t", 3); out += 3
            memcpy(out, lnk[2 * i + 1].p, lnk[2 * i + 1].n); out += lnk[2 * i + 1].n
            memcpy(out, b"\t+\t", 3); out += 3
            nd = snprintf(numbuf, 32, "%ld", ovl_c[i])
            memcpy(out, numbuf, nd); out += nd
            out[0] = b'M'; out[1] = b'\n'; out += 2
        if out - PyBytes_AS_STRING(text) != total:
            raise AssertionError("stage GFA size")
    finally:
        free(seg)
        free(lnk)
        free(ovl_c)
        free(new_of_old)
    return n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src[:n_e], a_tgt[:n_e], a_dp[:nv], text
