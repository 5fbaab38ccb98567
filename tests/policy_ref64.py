"""float64 evaluation of the DECLARED bi-head graph (agents/qlearnIA_V2.py:123-190 with the conventions of
oracle/policy_oracle.c) in torch CPU ops: the yardstick the fp32 paths are measured against.  Test infrastructure.

Not reference parity - keras / tensorflow and weights are absent from the image (parity of P1 stays UNPINNED) - but
an exact-arithmetic check of the restatement and of the HIP path: max |err| / max |value| per output tells how far
each fp32 evaluation order is from the graph itself, and the test tolerances are set from those numbers."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pyoracle


def _tensors(w, dtype):
    off, cnt, _ = pyoracle.policy_layout()
    cin = {"conv1": 2, "conv2": 8, "conv3": 8, "conv4": 8, "upconv1": 1, "upconv2": 2, "upconv3": 4, "upconv4": 8}
    cout = {"conv1": 8, "conv2": 8, "conv3": 8, "conv4": 8, "upconv1": 2, "upconv2": 4, "upconv3": 8, "upconv4": 1}
    dense = {"dense1": (5008, 100), "dense2": (100, 50), "output1": (50, 2), "updense1": (100, 625)}
    T = {}
    for name, o, c in zip(pyoracle.POLICY_TENSORS, off, cnt):
        layer, kind = name.split(".")
        v = torch.from_numpy(np.asarray(w[o:o + c], np.float64)).to(dtype)
        if kind == "kernel":
            v = v.reshape(dense[layer]) if layer in dense else v.reshape(3, 3, cin[layer], cout[layer])
        T[name] = v
    return T


def upsample2(u, legacy=False):
    """UpSampling2D((2, 2), interpolation='bilinear') on NCHW.  Half-pixel centres by default; legacy = the TF1
    resize_bilinear(align_corners=False) mapping src = dst / 2, written as an explicit gather (independent of the
    restatement's loop): out[2k] = in[k], out[2k+1] = (in[k] + in[min(k+1, n-1)]) / 2 along each axis."""
    if not legacy:
        return F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=False)
    n = u.shape[-1]
    k = torch.arange(2 * n)
    i0, i1 = k // 2, torch.clamp(k // 2 + 1, max=n - 1)
    w1 = ((k % 2) * 0.5).to(u.dtype)
    u = u[..., i0, :] * (1 - w1)[:, None] + u[..., i1, :] * w1[:, None]
    return u[..., :, i0] * (1 - w1) + u[..., :, i1] * w1


def forward(sm, lm, vec8s, w, dtype=torch.float64, legacy_bilinear=False):
    """One arena: maps (400,400) uint8, vec8s [K][8] heads of K ships -> act [K][2], heat [K][400][400] (numpy, dtype)."""
    T = _tensors(w, dtype)

    def conv(x, name):                              # HWIO -> OIHW, padding 'same'
        return F.conv2d(x, T[name + ".kernel"].permute(3, 2, 0, 1), T[name + ".bias"], padding=1)

    def bn(x, name):
        return F.batch_norm(x, T[name + ".mean"], T[name + ".var"], T[name + ".gamma"], T[name + ".beta"],
                            training=False, eps=1e-3)

    x = torch.from_numpy(np.stack([sm, lm], 0).astype(np.float64)).to(dtype)[None]   # NCHW
    for i in (1, 2, 3, 4):
        x = F.max_pool2d(F.relu(bn(conv(x, "conv%d" % i), "conv%d" % i)), 2)
    flat = x.permute(0, 2, 3, 1).reshape(1, -1)                                        # Flatten (h, w, c)
    v = torch.from_numpy(np.asarray(vec8s, np.float64)).to(dtype)
    cat = torch.cat([v, flat.expand(v.shape[0], -1)], 1)                               # vector first
    d1 = F.relu(cat @ T["dense1.kernel"] + T["dense1.bias"])
    d2 = F.relu(d1 @ T["dense2.kernel"] + T["dense2.bias"])
    act = d2 @ T["output1.kernel"] + T["output1.bias"]
    u = F.relu(d1 @ T["updense1.kernel"] + T["updense1.bias"]).reshape(-1, 1, 25, 25)
    for i in (1, 2, 3):
        u = upsample2(u, legacy_bilinear)
        u = F.relu(bn(conv(u, "upconv%d" % i), "upconv%d" % i))
    u = upsample2(u, legacy_bilinear)
    heat = conv(u, "upconv4")[:, 0]
    return act.numpy(), heat.numpy()


def frame_mask():
    m = np.zeros((400, 400), bool)
    m[0, :] = m[-1, :] = m[:, 0] = m[:, -1] = True
    return m


def errors(act, heat, act64, heat64):
    """max |err| / max |value| of the three outputs the tests bound: act, heat interior, heat frame."""
    fm = frame_mask()
    hs = float(np.abs(heat64).max())
    e = np.abs(heat.astype(np.float64) - heat64)
    return (float(np.abs(act.astype(np.float64) - act64).max() / max(1.0, np.abs(act64).max())),
            float(e[~fm].max() / hs), float(e[fm].max() / hs))
