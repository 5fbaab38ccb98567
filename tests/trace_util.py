"""Shared helpers to replay tests/golden/step_*.npz traces (recorded from the
live reference by oracle/gen_golden.py) through an engine and compare."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def step_traces():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "step_*.npz")))


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, "step_%s.npz" % name))
    return {k: z[k] for k in z.files}


def unpack_map(bits, w=400, h=400):
    return np.unpackbits(bits)[: w * h].reshape(w, h)
