"""Diagnostic (GPU box): lean and plain form of the textbook fit at `rows` rows, each against torch autograd in float64 on
the CPU (tests/test_train.py's checker) - which of the two forms drifts when they disagree.
usage: python tools/fit_check64.py [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import DeviceBuffer, _native as nat   # noqa: E402
from oracle import pyoracle                              # noqa: E402
from tests.test_train import _collect_minibatch, _fit_once, _torch_reference   # noqa: E402

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b, n, rows_d, bp_d, bn_d = _collect_minibatch(rows_n // 4)
w, shapes = pyoracle.policy_init(5, trained_like=True)
rs = np.random.RandomState(3)
y_act, y_ptr = rs.uniform(-1, 2, n).astype(np.float32), rs.uniform(-1, 2, n).astype(np.float32)
y, y2 = DeviceBuffer(4 * n).upload(y_act), DeviceBuffer(4 * n).upload(y_ptr)
bufs = tuple(DeviceBuffer(w.nbytes) for _ in range(4))
out = {}
for form in ("lean", "plain"):
    b.set_option(nat.OPT_FIT_PLAIN, int(form == "plain"))
    out[form] = _fit_once(b, "textbook", w, n, rows_d, bp_d, bn_d, y, y2, bufs)
rows = rows_d.download(b.TRANSITION_DTYPE, (n,))
bits = bp_d.download(np.uint32, (n, 2, 5000))
x0 = np.unpackbits(bits.view(np.uint8), bitorder="little").reshape(n, 2, 400, 400).astype(np.float64)
t0 = time.time()
import torch
torch.set_num_threads(16)
rl1, rl2, rg, stats = _torch_reference(w.astype(np.float64), shapes, x0, rows["head_prev"], rows["iaction"].astype(np.int64),
                                       rows["px"].astype(np.int64), rows["py"].astype(np.int64), y_act, y_ptr)
print("rows %d  float64 reference %.0f s  losses f64 %r lean %r plain %r" % (n, time.time() - t0, (rl1, rl2), out["lean"][0], out["plain"][0]))
for name, (o, shp) in shapes.items():
    c = int(np.prod(shp))
    if name.endswith((".mean", ".var")):
        continue
    ref = rg[o:o + c]
    scale = max(float(np.abs(ref).max()), 1e-30)
    el = float(np.abs(out["lean"][1][o:o + c] - ref).max()) / scale
    ep = float(np.abs(out["plain"][1][o:o + c] - ref).max()) / scale
    print("%-18s scale %.3e  lean %.2e  plain %.2e  %s" % (name, scale, el, ep, "<-- lean" if el > 1e-4 and el > ep else ("<-- plain" if ep > 1e-4 else "")))
b.close()
