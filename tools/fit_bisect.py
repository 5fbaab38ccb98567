"""Diagnostic (GPU box): where in a 1024-row minibatch do the lean and the plain form of the textbook fit disagree?
Runs both forms on windows [start, start + count) of the gathered rows and bisects the worst window.
usage: python tools/fit_bisect.py [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import DeviceBuffer, _native as nat   # noqa: E402
from oracle import pyoracle                              # noqa: E402
from tests.test_train import _collect_minibatch          # noqa: E402

rows_n = int(([a for a in sys.argv[1:] if not a.startswith('--')] or [1024])[0])
b, n, rows_d, bp_d, bn_d = _collect_minibatch(rows_n // 4)
w, shapes = pyoracle.policy_init(5, trained_like=True)
rs = np.random.RandomState(3)
y = DeviceBuffer(4 * n).upload(rs.uniform(-1, 2, n).astype(np.float32))
y2 = DeviceBuffer(4 * n).upload(rs.uniform(-1, 2, n).astype(np.float32))
w_d, m_d, v_d, g_d = (DeviceBuffer(w.nbytes) for _ in range(4))
zeros = np.zeros_like(w)
ROW = b.TRANSITION_DTYPE.itemsize


def run(form, start, count):
    b.set_option(nat.OPT_FIT_PLAIN, int(form == "plain"))
    w_d.upload(w); m_d.upload(zeros); v_d.upload(zeros)
    b.dqn_fit(w_d, m_d, v_d, 1, 1e-4, count, rows_d.ptr + start * ROW, bp_d.ptr + start * 40000, y.ptr + 4 * start,
              y2.ptr + 4 * start, g_d)
    return g_d.download(np.float32, w.shape).astype(np.float64)


def diff(start, count, names=("conv1.kernel", "conv1.gamma", "conv2.kernel")):
    ga, gb = run("lean", start, count), run("plain", start, count)
    out = []
    for name in names:
        o, shp = shapes[name]
        c = int(np.prod(shp))
        out.append(float(np.abs(ga[o:o + c] - gb[o:o + c]).max() / max(np.abs(gb[o:o + c]).max(), 1e-30)))
    return out


print("whole", n, diff(0, n))
if "--shift" in sys.argv:
    for start in (0, 1, 4, 512, 1000, 1024):
        if start + 1024 <= n:
            print("window", start, 1024, ["%.2e" % v for v in diff(start, 1024)])
    b.close()
    sys.exit(0)
if "--whole" in sys.argv:
    for count in (1023, 1024, 512, 2047 if n >= 2048 else 1000, 2048 if n >= 2048 else 1008):
        print("prefix", count, ["%.2e" % v for v in diff(0, count)])
    b.close()
    sys.exit(0)
if "--prefix" in sys.argv:
    for count in (520, 544, 576, 640, 704, 768, 832, 896, 960, 1000, 1023):
        print("prefix", count, ["%.2e" % v for v in diff(0, count)], " suffix", ["%.2e" % v for v in diff(n - count, count)])
    b.close()
    sys.exit(0)
for count in (512, 256, 128):
    for start in range(0, n, count):
        print("window", start, count, ["%.2e" % v for v in diff(start, count)])
lo, cnt = 0, n
while cnt > 1:
    h = cnt // 2
    a, c = diff(lo, h)[0], diff(lo + h, cnt - h)[0]
    print("bisect [%d, %d): %.2e   [%d, %d): %.2e" % (lo, lo + h, a, lo + h, lo + cnt, c))
    if c > a:
        lo += h
        cnt -= h
    else:
        cnt = h
print("worst single row", lo, diff(lo, 1))
rows = rows_d.download(b.TRANSITION_DTYPE, (n,))
print(rows[lo])
b.close()
