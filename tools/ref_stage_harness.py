#!/usr/bin/env python3
"""Run the REAL reference command (/root/reference/vstrains) in this process behind tests/golden/gt_standin and
time the interval BASELINE.json's second metric half names -- "pe_info available -> strain_dict returned", i.e.
everything utils/VStrains_SPAdes.py:133-272 does after the PE subprocess returned -- as ONE measured interval
(perf_counter at the return of subprocess.check_call, perf_counter when VStrains_SPAdes.run returns), not as a
difference of two clocks.  Build container only (the reference cannot travel); launched by
tools/time_reference_stages.py, which sets PYTHONPATH (stand-in first), PYTHONHASHSEED and GT_STANDIN_INEDGE.

    python tools/ref_stage_harness.py --timing out.json [--pe-cache DIR] -- <argv of the reference command>

--pe-cache DIR: DIR/pe_info, DIR/st_info and DIR/s_graph_L1.sha256 were written by the real
VStrains_PE_Inference.py on a byte-identical s_graph_L1.gfa (the harness checks the SHA-256 of the file the
reference has just written; another s_graph_L1.gfa -- the stand-in's other in-edge model numbers the segments differently
-- gets a cache of its own, made by the real script on first use); the PE subprocess -- whose output depends on the graph
file and the reads only -- is then replaced by a copy of those files, so that eight runs need two PE runs.

Every INFO line the reference logs is appended to <timing>.progress with the seconds since start, as it happens:
a run that is killed leaves how far it got and at which rate.
"""
import hashlib
import json
import logging
import os
import runpy
import shutil
import sys
import time

REF = "/root/reference"


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    mine, theirs = argv[:cut], argv[cut + 1:]
    timing_path = mine[mine.index("--timing") + 1]
    pe_cache = mine[mine.index("--pe-cache") + 1] if "--pe-cache" in mine else None
    sys.path.insert(0, REF)
    t_start = time.perf_counter()
    marks = {"milestones": []}
    progress = open(timing_path + ".progress", "w")

    orig_info = logging.Logger.info

    def info(self, msg, *a, **kw):
        now = time.perf_counter() - t_start
        text = str(msg)
        progress.write("%.3f\t%s\n" % (now, text[:200].replace("\n", " ")))
        progress.flush()
        if text.startswith(">>>STAGE") or text in ("paired end information stored", "VStrains-SPAdes finished", "VStrains-SPAdes started"):
            marks["milestones"].append([round(now, 4), text])
        return orig_info(self, msg, *a, **kw)

    logging.Logger.info = info

    from utils import VStrains_SPAdes as drv

    real_check_call = drv.subprocess.check_call

    class _Sub:
        """the module's `subprocess` name, check_call instrumented; everything else is the real module's"""

        def __getattr__(self, name):
            return getattr(sys.modules["subprocess"], name)

        @staticmethod
        def check_call(cmd, **kw):
            marks["pe_call_at_s"] = time.perf_counter() - t_start
            if pe_cache is None:
                real_check_call(cmd, **kw)
                marks["pe_subprocess"] = "run"
            else:
                words = cmd.split()
                gfa, out = words[words.index("-g") + 1], words[words.index("-o") + 1]
                with open(gfa, "rb") as fh:
                    sha = hashlib.sha256(fh.read()).hexdigest()
                with open(os.path.join(pe_cache, "s_graph_L1.sha256")) as fh:
                    want = fh.read().strip()
                cache = pe_cache
                if sha != want:
                    # another s_graph_L1.gfa (the stand-in's other in-edge model walks the input graph in another order and
                    # numbers / orients the segments differently): a cache of its own, made by the real script on first use
                    cache = pe_cache + "_" + sha[:16]
                    if not os.path.exists(os.path.join(cache, "s_graph_L1.sha256")):
                        real_check_call(cmd, **kw)
                        tmp = cache + ".tmp%d" % os.getpid()
                        os.makedirs(tmp)
                        for name in ("pe_info", "st_info"):
                            shutil.copyfile(os.path.join(out, name), os.path.join(tmp, name))
                        with open(os.path.join(tmp, "s_graph_L1.sha256"), "w") as fh:
                            fh.write(sha + "\n")
                        try:
                            os.rename(tmp, cache)
                        except OSError:
                            shutil.rmtree(tmp)
                        marks["pe_subprocess"] = "run (first use of this s_graph_L1.gfa)"
                        marks["pe_returned_at_s"] = time.perf_counter() - t_start
                        marks["_t_pe"] = time.perf_counter()
                        return
                if os.path.exists(out):  # PE_Inference.py:93-96 deletes and recreates -o
                    shutil.rmtree(out)
                os.makedirs(out)
                for name in ("pe_info", "st_info"):
                    shutil.copyfile(os.path.join(cache, name), os.path.join(out, name))
                marks["pe_subprocess"] = "replaced by the files the real script wrote on the same s_graph_L1.gfa (sha256 checked)"
            marks["pe_returned_at_s"] = time.perf_counter() - t_start
            marks["_t_pe"] = time.perf_counter()

    drv.subprocess = _Sub()
    real_run = drv.run

    def run(args, logger):
        try:
            return real_run(args, logger)
        finally:
            t = time.perf_counter()
            marks["run_returned_at_s"] = t - t_start
            if "_t_pe" in marks:
                marks["after_pe_to_run_return_s"] = t - marks.pop("_t_pe")
            marks["exception"] = repr(sys.exc_info()[1]) if sys.exc_info()[1] is not None else None
            with open(timing_path, "w") as fh:
                json.dump(marks, fh, indent=1)

    drv.run = run
    sys.argv = [os.path.join(REF, "vstrains")] + theirs
    runpy.run_path(os.path.join(REF, "vstrains"), run_name="__main__")


if __name__ == "__main__":
    main()
