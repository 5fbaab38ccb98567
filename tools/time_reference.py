#!/opt/conda/bin/python3.9
"""Times the LIVE reference's tick loop - Battleground.frame() (lib/battleground.py:163-166) with the GUI's laser
clean-up between ticks (lib/ofighters.py:619-625,702-707) and Battleground.restart() every 200 ticks (:684-688) - in
the BUILD CONTAINER only (the reference never travels to the GPU box).  Same import shim as oracle/gen_golden.py:
keras / tensorflow are absent from the image and replaced by inert mocks for the import (scripted bots only; nothing
of the mocks is on the timed path).  BASELINE.md section 2 records the output.

Usage:  /opt/conda/bin/python3.9 tools/time_reference.py [--procs P]
  config 1 of BASELINE.json: 1 arena x 4 random ships x 200 iterations; and the 8-ship arena of the headline shape.
  One Python process = one core; --procs P runs P independent copies for the container's aggregate."""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
from unittest.mock import MagicMock

for _m in ["keras", "keras.models", "keras.layers", "keras.layers.core", "keras.optimizers",
           "keras.layers.advanced_activations", "keras.backend", "tensorflow"]:
    sys.modules[_m] = MagicMock()
sys.path.insert(0, "/root/reference")

import random

_sink = io.StringIO()
with contextlib.redirect_stdout(_sink):
    from ofighters.lib.battleground import Battleground


def episode(n_ships, ticks, seed):
    """arena-steps/s of `ticks` ticks of one arena of random bots (stdout of the reference suppressed)."""
    random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        bg = Battleground(ships={"random": n_ships})
        peak = 0
        t0 = time.perf_counter()
        for t in range(ticks):
            bg.frame()
            peak = max(peak, len(bg.lasers))
            bg.lasers = [l for l in bg.lasers if l.state != "destroyed"]   # the GUI's clear_wreckage
        dt = time.perf_counter() - t0
    return ticks / dt, peak


def best_of(n_ships, ticks=200, reps=5, seed=7):
    runs = [episode(n_ships, ticks, seed) for _ in range(reps)]
    return max(r[0] for r in runs), max(r[1] for r in runs)


def _worker(args):
    return best_of(*args)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    import numpy, skimage
    out = {"interpreter": platform.python_version(), "numpy": numpy.__version__, "scikit-image": skimage.__version__,
           "cores_in_container": os.cpu_count(), "unit": "arena-steps/s", "runs": []}
    for n_ships in (4, 8):
        one, peak = best_of(n_ships)
        with mp.Pool(a.procs) as pool:
            agg = sum(pool.map(_worker, [(n_ships, 200, 3, 7 + i) for i in range(a.procs)]))
        out["runs"].append({"workload": "1 arena x %d random ships x 200 ticks (step + obs build + random bot)" % n_ships,
                            "one_process_best_of_5": round(one, 1), "peak_live_lasers": peak,
                            "aggregate_%d_processes" % a.procs: round(agg, 1)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
