"""Diagnostic (GPU box): where does ofx_dqn_fit_reference's updense1 gradient error (1.25e-3 of its scale against float64) come from?
The same graph + dense targets through (a) torch float64 on the CPU (the checker), (b) torch float32 on the CPU (whose
BatchNorm backward accumulates and evaluates per element in double: at::acc_type<float> on the CPU is double),
(c) torch float32 on the GPU (float arithmetic throughout), (d) libofx.  Prints the error of updense1.{kernel,bias}
against (a) for each, and how the error of (d) is distributed over the 25 x 25 cells of u0.
usage (GPU box): python tools/fit_precision.py"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--torch-gpu":
    import torch                                   # torch's HIP runtime first (its own copy)
    from tests.test_train import _torch_reference
    z = np.load(sys.argv[2], allow_pickle=False)
    shapes = {k: (int(o), tuple(int(v) for v in s.split("x") if v)) for k, o, s in zip(z["names"], z["offs"], z["shps"])}
    # run the reference on the GPU by making new tensors land there
    torch.set_default_device("cuda")
    l1, l2, g, _ = _torch_reference(z["w"], shapes, z["xn"], z["head_next"], None, None, None, None, None,
                                    dense=(z["t1"], z["t2"]), dtype=torch.float32)
    np.save(sys.argv[3], g)
    sys.exit(0)

from ofighters_amd import ArenaBatch, DeviceBuffer
from oracle import pyoracle
from tests import policy_ref64 as R
from tests.test_train import _torch_reference

N, M, seed, batch, lr, gamma = 2, 4, 0x0F160001, 2, 1e-4, 0.9
b = ArenaBatch(N, M)
b.replay_create(16, 0)
b.spawn_random(seed)
w, shapes = pyoracle.policy_init(9, trained_like=True)
mask = np.zeros((N, M), np.uint8); mask[:, [1, 3]] = 1
mask_d = DeviceBuffer(mask.nbytes).upload(mask)
ia_d, ip_d = DeviceBuffer(4 * N * M), DeviceBuffer(8 * N * M)
for t in range(12):
    b.bot_actions(["random"] * M, seed, tick=t)
    b.policy_explore(1.0, seed, tick=t, collecting=True, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
    b.policy_actions(out_ptr=b._actions.ptr, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
    b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
    b.step(actions_ptr=b._actions.ptr)
slot, _ = b.replay_sample(5, 0, batch)
rows_d, bp_d, bn_d = b.replay_gather_device(slot, batch)
n = N * batch
rows = rows_d.download(b.TRANSITION_DTYPE, (n,))
w_d = DeviceBuffer(w.nbytes).upload(w)
zeros = np.zeros_like(w)
m_d, v_d, g_d = DeviceBuffer(w.nbytes).upload(zeros), DeviceBuffer(w.nbytes).upload(zeros), DeviceBuffer(w.nbytes)
b.dqn_fit_reference(w_d, m_d, v_d, 1, lr, n, rows_d.ptr, bp_d.ptr, bn_d.ptr, gamma, g_d)
g = g_d.download(np.float32, w.shape).astype(np.float64)

def maps(buf):
    bits = buf.download(np.uint32, (n, 2, 5000))
    return np.unpackbits(bits.view(np.uint8), bitorder="little").reshape(n, 2, 400, 400)
xp, xn = maps(bp_d), maps(bn_d)
t1, t2 = np.zeros((n, 2)), np.zeros((n, 400, 400))
for s in range(n):
    a_prev, h_prev = R.forward(xp[s, 0], xp[s, 1], rows["head_prev"][s][None], w)
    a_next, h_next = R.forward(xn[s, 0], xn[s, 1], rows["head_next"][s][None], w)
    t1[s], t2[s] = a_prev[0], h_prev[0]
    live = 0.0 if rows["done"][s] else 1.0
    t1[s, int(rows["iaction"][s] != 0)] = rows["reward"][s] + gamma * a_next[0].max() * live
    t2[s, rows["px"][s], rows["py"][s]] = rows["reward"][s] + gamma * h_next[0].max() * live
b.close()
import torch
args = (w.astype(np.float64), shapes, xn.astype(np.float64), rows["head_next"], None, None, None, None, None)
_, _, g64, _ = _torch_reference(*args, dense=(t1, t2))
_, _, g32, _ = _torch_reference(*args, dense=(t1, t2), dtype=torch.float32)
ggpu = None
with tempfile.TemporaryDirectory() as d:
    names = list(shapes)
    np.savez(d + "/in.npz", w=w.astype(np.float64), xn=xn.astype(np.float64), head_next=rows["head_next"], t1=t1, t2=t2,
             names=np.array(names), offs=np.array([shapes[k][0] for k in names]),
             shps=np.array(["x".join(str(v) for v in shapes[k][1]) for k in names]))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-gpu", d + "/in.npz", d + "/g.npy"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    if p.returncode == 0:
        ggpu = np.load(d + "/g.npy")
    else:
        print("torch GPU leg failed:", p.stderr[-600:])
for name in ("updense1.kernel", "updense1.bias", "upconv1.kernel", "dense1.kernel"):
    o, shp = shapes[name]; c = int(np.prod(shp))
    ref = g64[o:o + c]; sc = np.abs(ref).max()
    line = "%-16s scale %.3e | err/scale: torch cpu fp32 %.2e" % (name, sc, np.abs(g32[o:o + c] - ref).max() / sc)
    if ggpu is not None:
        line += " | torch gpu fp32 %.2e" % (np.abs(ggpu[o:o + c] - ref).max() / sc)
    line += " | libofx %.2e" % (np.abs(g[o:o + c] - ref).max() / sc)
    print(line)
o, shp = shapes["updense1.bias"]
err = (g[o:o + 625] - g64[o:o + 625]).reshape(25, 25); sc = np.abs(g64[o:o + 625]).max()
print("updense1.bias error map / scale: rms %.2e, cells above 1e-4: %d of 625, worst cells (y, x, err):" %
      (np.sqrt((err ** 2).mean()) / sc, int((np.abs(err) > 1e-4 * sc).sum())))
for k in np.argsort(-np.abs(err).ravel())[:6]:
    print("   ", k // 25, k % 25, "%.2e" % (err.ravel()[k] / sc))
