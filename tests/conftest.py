import os
import sys

import pytest

# (the 'untouched' hints of the graph stages are verified against the snapshot in every test run)
os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")
# (No VS_EXPERIMENT here: the suite runs the library as a user gets it.  The parity-safe tuning switches of vs_pe_count
# -- VS_NO_SORT, VS_EPT, ... -- exist only in a context created with VS_EXPERIMENT=1; the variant tests ask for one
# through ``experiment_context`` and flip the switches on it.)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pe_cases(ok_only=True):
    import json

    base = os.path.join(GOLDEN, "pe")
    out = []
    for name in sorted(os.listdir(base)):
        d = os.path.join(base, name)
        if not os.path.isdir(d):
            continue
        with open(os.path.join(d, "meta.json")) as fh:
            meta = json.load(fh)
        if ok_only and meta["returncode"] != 0:
            continue
        out.append((name, d, meta))
    return out


def experiment_context(host, device=0):
    """A context made in experiment mode (VsTuning level 1): the switches are read from the environment on every call."""
    old = os.environ.get("VS_EXPERIMENT")
    os.environ["VS_EXPERIMENT"] = "1"
    try:
        return host.Context(device)
    finally:
        if old is None:
            os.environ.pop("VS_EXPERIMENT", None)
        else:
            os.environ["VS_EXPERIMENT"] = old


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
