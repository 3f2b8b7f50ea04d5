#!/usr/bin/env python3
"""Build container only: how long a fork chain the REAL reference's ``contig_dict_remapping`` can follow.

``merge_id`` (Utilities.py:318-327) recurses once per link of a chain of forked ids; the reference's CLI calls
``contig_dict_remapping`` six Python frames deep (<module>, main, run, VStrains_SPAdes.run, path_extension or
iter_graph_disentanglement, contig_dict_remapping itself), under the interpreter's default recursion limit.  This script
imports the function from /root/reference (behind tests/golden/gt_standin, like every reference run here), calls it at
exactly that depth on synthetic chains "n0" -> "n1" -> ... of growing length, and records the longest chain it returns
from and the exception the next one raises.  ``tests/golden/merge_id_depth.json`` is what the native stage engine
(csrc/vs_stage.cpp: PY_MERGE_ID_FRAMES) and the Python checker (oracle/graph_stages/contig_ops.py) are held to.

    python tests/golden/probe_merge_id_depth.py
"""
import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gt_standin"))
sys.path.insert(0, "/root/reference")


def frame2(fn, *a):  # <module> is frame 1; these stand for main, run, VStrains_SPAdes.run, path_extension
    return frame3(fn, *a)


def frame3(fn, *a):
    return frame4(fn, *a)


def frame4(fn, *a):
    return frame5(fn, *a)


def frame5(fn, *a):
    return fn(*a)  # contig_dict_remapping = frame 6


def chain(frames):
    """a chain whose deepest merge_id call is nested ``frames`` deep: n0 -> n1 -> ... -> n(frames-1), the last not forked"""
    ids = ["n%d" % i for i in range(frames)]
    id_mapping = {ids[i]: {ids[i + 1]} for i in range(frames - 1)}
    id_mapping[ids[-1]] = set()
    return ids, id_mapping


if __name__ == "__main__":
    from utils.VStrains_Utilities import contig_dict_remapping

    logger = logging.getLogger("probe")
    logger.handlers[:] = [logging.NullHandler()]
    logger.propagate = False
    # (the calls are made from module level, so that frame2..frame5 + the function are frames 2..6 as under the CLI)
    lo, hi, error = 1, 5000, None
    while hi - lo > 1:
        mid = (lo + hi) // 2
        ids, id_mapping = chain(mid)
        try:
            red = frame2(contig_dict_remapping, None, {}, {}, {}, id_mapping, [ids[0]], logger)
            assert red[ids[0]] == {ids[-1]}
            lo = mid
        except RecursionError as err:
            hi, error = mid, "RecursionError: %s" % err
    out = {"max_nested_merge_id_frames": lo, "error_one_beyond": error, "python": sys.version.split()[0],
           "recursion_limit": sys.getrecursionlimit(), "call_depth_of_contig_dict_remapping": 6,
           "produced_by": "contig_dict_remapping imported from /root/reference/utils/VStrains_Utilities.py (:281-380), "
                          "tests/golden/probe_merge_id_depth.py in the build container"}
    with open(os.path.join(HERE, "merge_id_depth.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(out))
