"""Helpers shared by the graph-stage tests: load a golden case, run the pipeline, collect the
outputs in the same digest form ``tests/golden/make_graph_golden.py`` stored."""
import argparse
import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_graph_golden as gold  # noqa: E402  (digest helpers + case table; runs nothing on import)
from vstrains_amd import synth  # noqa: E402

GOLDEN = os.path.join(HERE, "golden", "graph")


def case_names():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "case.json")))


class Case:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        with open(os.path.join(self.dir, "case.json")) as fh:
            self.meta = json.load(fh)
        self.expected = {}
        for rel in self.meta["files"]:
            with open(os.path.join(self.dir, "out", rel)) as fh:
                self.expected[rel] = fh.read()

    # Files of this case that do not pin the build to the reference: the reference itself writes them differently under
    # another PYTHONHASHSEED (it iterates sets of names), or the stand-in writes them differently under another of its
    # models of graph-tool's adjacency container (in-edge placement, edge-index reuse, remove_edge: the rules recalled,
    # not read -- tests/golden/gt_standin).  They are still compared and reported; only the others are binding.
    MODEL_FIELDS = ("differs_under_plain_inedge_order", "differs_under_lifo_index_reuse", "differs_under_swap_pop_removal")

    def model_dependent(self):
        dep = set()
        for field in self.MODEL_FIELDS:
            dep.update(self.meta.get(field, []))
        return dep

    def non_binding(self):
        return set(self.meta["differs_under_other_hashseeds"]) | self.model_dependent()

    def binding(self, problems, report=None):
        """``problems`` of ``compare`` without the ones in non-binding files (those go to ``report``, if given)."""
        skip = self.non_binding()
        out = []
        for p in problems:
            rel = p.split(" ", 1)[1]
            if rel in skip:
                if report is not None:
                    report.append("%s: %s (non-binding)" % (self.name, p))
            else:
                out.append(p)
        return out

    def expected_exception(self):
        """The exception the reference ended with on this case (``returncode`` != 0), named by the last line of its
        traceback: KeyError from its PE subprocess (a node base outside ACGT), RecursionError from merge_id after a runaway
        trivial split (Utilities.py:318-327) -- this build must raise the same one after writing the same files."""
        import builtins

        assert self.meta["returncode"] != 0
        last = [l for l in self.meta.get("stderr_tail", "").strip().splitlines() if l.strip()][-1]
        name = last.split(":", 1)[0].strip()
        if name == "subprocess.CalledProcessError":  # (the PE script, a subprocess of the reference, died of the KeyError)
            name = "KeyError"
        exc = getattr(builtins, name, None)
        assert isinstance(exc, type) and issubclass(exc, BaseException), last
        return exc

    def inputs(self, tmp, with_reads=False):
        """Re-derive the inputs from the seed and check them against the pinned digests."""
        pc = synth.make_pipeline_case(**self.meta["synth"])
        assert gold.md5(pc.gfa_text) == self.meta["input_md5"]["gfa"], "synthetic graph drifted"
        assert gold.md5(pc.paths_text) == self.meta["input_md5"]["paths"], "synthetic contigs drifted"
        paths = {"gfa": os.path.join(self.dir, "in", "graph.gfa"), "paths": os.path.join(self.dir, "in", "contigs.paths")}
        if with_reads:
            ft, rt = synth.fastq_text(pc.fwd, "f"), synth.fastq_text(pc.rve, "r")
            assert gold.md5(ft) == self.meta["input_md5"]["fwd"], "synthetic reads drifted"
            assert gold.md5(rt) == self.meta["input_md5"]["rve"]
            for key, text in (("fwd", ft), ("rve", rt)):
                paths[key] = os.path.join(tmp, key + ".fq")
                with open(paths[key], "w") as fh:
                    fh.write(text)
        else:
            paths["fwd"] = paths["rve"] = os.path.join(tmp, "unused.fq")
        return paths

    def args(self, inp, out):
        extra = self.meta["cli_extra"]
        min_cov = int(extra[extra.index("-mc") + 1]) if "-mc" in extra else None
        min_len = int(extra[extra.index("-ml") + 1]) if "-ml" in extra else 250
        for sub in ("gfa", "tmp", "paf", "aln"):
            os.makedirs(os.path.join(out, sub))
        return argparse.Namespace(gfa_file=inp["gfa"], path_file=inp["paths"], fwd=inp["fwd"], rve=inp["rve"],
                                  output_dir=out, min_cov=min_cov, min_len=min_len, ref_file=None, dev=False)

    def write_info_files(self, names, aln_dir):
        """pe_info / st_info text rebuilt from the stored non-zero lines."""
        os.makedirs(aln_dir, exist_ok=True)
        for fname in ("pe_info", "st_info"):
            lines = self.expected["aln/" + fname].split("\n")
            total = int(lines[0])
            assert total == len(names) ** 2
            nz = {}
            for l in lines[1:]:
                if l:
                    u, v, c = l.split(":")
                    nz[(u, v)] = c
            with open(os.path.join(aln_dir, fname), "w") as fh:
                for u in names:
                    fh.write("".join("%s:%s:%s\n" % (u, v, nz.get((u, v), "0")) for v in names))


def quiet_logger(name="vstrains-test"):
    lg = logging.getLogger(name)
    lg.handlers[:] = [logging.NullHandler()]
    lg.setLevel(logging.CRITICAL)
    lg.propagate = False
    return lg


def file_logger(out_dir, name="vstrains-test-file"):
    """Logger that writes OUT/vstrains.log the way the reference's CLI does while its pipeline runs
    (vstrains:245-247: "%(asctime)s - %(levelname)s - %(message)s", DEBUG level as under -d), so that
    the INFO lines can be compared with the reference's (make_graph_golden.info_log_lines)."""
    lg = logging.getLogger(name)
    for h in list(lg.handlers):
        lg.removeHandler(h)
        h.close()
    fh = logging.FileHandler(os.path.join(out_dir, "vstrains.log"), mode="w")
    fh.setLevel(logging.DEBUG)
    fh.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
    lg.addHandler(fh)
    lg.setLevel(logging.DEBUG)
    lg.propagate = False
    return lg


def compare(case, out_dir, skip=()):
    got = gold.collect(out_dir)
    problems = []
    for rel in case.meta["files"]:
        if rel in skip:
            continue
        if rel == "vstrains.log.info" and rel not in got:
            continue  # (this run kept no log file: nothing to compare the reference's INFO lines with)
        if rel not in got:
            problems.append("missing " + rel)
        elif got[rel] != case.expected[rel]:
            problems.append("differs " + rel)
    extra = sorted(set(got) - set(case.meta["files"]))
    if extra:
        problems.append("unexpected files " + ",".join(extra))
    return problems, got


def reference_command_inputs(work, config=0):
    """The inputs the REAL reference command was given at BASELINE configs[config] (and the digests committed for it,
    tests/golden/reference_digests.json "configs[i]_whole_command"; tools/time_reference.py for configs[0],
    tools/time_reference_stages.py for configs[1] and configs[2]): the config's assembler-style GFA + contig paths and the
    first ``pairs`` read pairs of its bench stream as FASTQ text (the CPU twin of the device generator, the same stream
    the bench counts)."""
    import json

    from oracle import pe_oracle_c
    from vstrains_amd.workloads import CONFIGS, workload_for

    with open(os.path.join(HERE, "golden", "reference_digests.json")) as fh:
        want = json.load(fh)["configs[%d]_whole_command" % config]
    cfg = CONFIGS[config]
    st, pre, names, seqs, cum, logger, _ = workload_for(config, work)
    L = cfg["read_len"]
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, want["stream_seed"], 0, want["pairs"], L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    paths = {"gfa": os.path.join(work, "input.gfa"), "paths": os.path.join(work, "input.paths")}
    qual = b"I" * L
    for key, arr, tag in (("fwd", fw, b"f"), ("rve", rv, b"r")):
        paths[key] = os.path.join(work, key + ".fq")
        with open(paths[key], "wb") as fh:
            for lo in range(0, arr.shape[0], 100000):
                fh.write(b"".join(b"@%s%d\n%s\n+\n%s\n" % (tag, i, arr[i].tobytes(), qual) for i in range(lo, min(arr.shape[0], lo + 100000))))
    return paths, want


def strain_set_sha256(fasta_path):
    """The extracted strain sequences as a set, each on its lexicographically smaller strand (tools/time_reference_stages.py
    writes the same digest of the REAL reference's strain.fasta per run: at configs[0] and configs[2] it is the same under
    BOTH in-edge models of the stand-in -- the one statement about the final result that does not depend on the model)."""
    import hashlib

    comp = str.maketrans("ACGT", "TGCA")
    seqs = []
    with open(fasta_path) as fh:
        for line in fh:
            line = line.strip()
            if line and not line.startswith(">"):
                seqs.append(min(line, line[::-1].translate(comp)))
    return hashlib.sha256("\n".join(sorted(seqs)).encode()).hexdigest()


def reference_command_problems(out_dir, want):
    """Files of ``want["files_sha256"]`` (the ones all eight runs of the real command agree on: both in-edge models of the
    stand-in, hash seeds 0-3) that ``out_dir`` does not reproduce."""
    import hashlib

    got = gold.collect(out_dir)
    problems = ["%s %s" % ("missing" if rel not in got else "differs", rel) for rel, sha in sorted(want["files_sha256"].items())
                if rel not in got or hashlib.sha256(got[rel].encode()).hexdigest() != sha]
    if "strain_set_sha256" in want and strain_set_sha256(os.path.join(out_dir, "strain.fasta")) != want["strain_set_sha256"]:
        problems.append("differs the set of strain sequences (strand-normalised)")
    return problems
