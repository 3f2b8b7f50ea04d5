"""Section timers for the graph stages (diagnostics: ``VS_STAGE_TIMING=1``; otherwise ``ON`` is False and the callers skip
the bookkeeping).  ``bench.py --extract`` reports ``SECTIONS`` next to the stage times."""
import os
import time

ON = os.environ.get("VS_STAGE_TIMING", "") not in ("", "0")
SECTIONS = {}
now = time.perf_counter


def add(name: str, t0: float) -> float:
    t = now()
    SECTIONS[name] = SECTIONS.get(name, 0.0) + (t - t0)
    return t
