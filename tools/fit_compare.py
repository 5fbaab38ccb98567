"""Diagnostic (GPU box): the lean and the plain form of the fit on one minibatch of `rows` rows - per-tensor difference
of the gradients, the losses.  usage: python tools/fit_compare.py [rows] [--reference]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
from oracle import pyoracle

args = sys.argv[1:]
quirks = "--reference" in args
n_want = ([int(a) for a in args if not a.startswith("--")] or [256])[0]
M, batch, seed = 4, 4, 0x0F160077
N = (n_want + batch - 1) // batch
b = ArenaBatch(N, M)
b.replay_create(16, 0)
b.spawn_random(seed)
w, shapes = pyoracle.policy_init(5, trained_like=True)
mask = np.zeros((N, M), np.uint8); mask[:, [0, 3]] = 1
mask_d = DeviceBuffer(mask.nbytes).upload(mask)
ia_d, ip_d = DeviceBuffer(4 * N * M), DeviceBuffer(8 * N * M)
for t in range(10):
    b.bot_actions(["random"] * M, seed, tick=t)
    b.policy_explore(1.0, seed, tick=t, collecting=True, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
    b.policy_actions(out_ptr=b._actions.ptr, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
    b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
    b.step(actions_ptr=b._actions.ptr)
slot, _ = b.replay_sample(7, 0, batch)
rows_d, bp_d, bn_d = b.replay_gather_device(slot, batch)
n = N * batch
rs = np.random.RandomState(3)
y = DeviceBuffer(4 * n).upload(rs.uniform(-1, 2, n).astype(np.float32))
y2 = DeviceBuffer(4 * n).upload(rs.uniform(-1, 2, n).astype(np.float32))
zeros = np.zeros_like(w)
w_d, m_d, v_d, g_d = (DeviceBuffer(w.nbytes) for _ in range(4))
out = {}
for form in ("lean", "plain"):
    b.set_option(nat.OPT_FIT_PLAIN, int(form == "plain"))
    w_d.upload(w); m_d.upload(zeros); v_d.upload(zeros)
    if quirks:
        l = b.dqn_fit_reference(w_d, m_d, v_d, 1, 1e-4, n, rows_d.ptr, bp_d.ptr, bn_d.ptr, 0.9, g_d)
    else:
        l = b.dqn_fit(w_d, m_d, v_d, 1, 1e-4, n, rows_d.ptr, bp_d.ptr, y.ptr, y2.ptr, g_d)
    out[form] = (l, g_d.download(np.float32, w.shape).astype(np.float64), w_d.download(np.float32, w.shape))
print("rows", n, "losses lean", out["lean"][0], "plain", out["plain"][0])
ga, gb = out["lean"][1], out["plain"][1]
for name, (o, shp) in shapes.items():
    c = int(np.prod(shp))
    if name.endswith((".mean", ".var")):
        d = np.abs(out["lean"][2][o:o + c] - out["plain"][2][o:o + c]).max()
        print("%-18s moved statistics differ by %.2e" % (name, d))
        continue
    scale = np.abs(gb[o:o + c]).max()
    print("%-18s scale %.3e  max diff %.2e of scale" % (name, scale, np.abs(ga[o:o + c] - gb[o:o + c]).max() / max(scale, 1e-30)))
b.close()
