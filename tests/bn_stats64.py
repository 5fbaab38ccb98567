"""float64 checker for the BatchNorm batch statistics of a big minibatch (tests/test_train.py::test_lean_fit_at_4096_rows):
the training-mode forward of the declared graph (agents/qlearnIA_V2.py:123-190 with Keras' training=True) in torch
float64 on the GPU, layer by layer over the whole minibatch.  Runs as a process of its own (python tests/bn_stats64.py
in.npz out.npz); imports nothing of libofx."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def conv3x3_f64(x, k, bias):
    """x [n][h][w][ci] float64 (torch, any device), k HWIO, zero padding 1: nine shifted matmuls - no library convolution
    (MIOpen has no float64 path), nothing of libofx."""
    import torch
    import torch.nn.functional as F
    n, h, w_, ci = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    z = bias.reshape(1, 1, 1, -1).expand(n, h, w_, k.shape[3]).clone()
    for ky in range(3):
        for kx in range(3):
            z += xp[:, ky:ky + h, kx:kx + w_, :] @ k[ky, kx]
    return z


def batch_statistics_f64(w, shapes, bits, vec8, chunk=256, device="cuda"):
    """The BatchNorm batch statistics (mean, biased variance) of every normalised convolution of the training-mode
    forward (qlearnIA_V2.py:123-190 with Keras' training=True), float64 torch on the GPU, layer by layer over the whole
    minibatch - the checker for the moved statistics of a 4096-row fit."""
    import torch
    from tests.policy_ref64 import upsample2
    dev = torch.device(device)
    f64 = torch.float64
    P = {name: torch.tensor(w[o:o + int(np.prod(shp))].reshape(shp), dtype=f64, device=dev) for name, (o, shp) in shapes.items()}
    n = bits.shape[0]
    stats = {}

    def bn_block(xs, name, pre, post=lambda x: x):
        """xs: list of chunk inputs (NHWC); pre / post: per-chunk transforms in front of the convolution / behind the
        ReLU.  Two passes: statistics, then post(relu(bn(z))) per chunk."""
        s1 = torch.zeros(P[name + ".bias"].shape, dtype=f64, device=dev)
        s2 = torch.zeros_like(s1)
        cnt = 0
        zs = []
        for x in xs:
            z = conv3x3_f64(pre(x), P[name + ".kernel"], P[name + ".bias"])
            s1 += z.sum(dim=(0, 1, 2)); s2 += (z * z).sum(dim=(0, 1, 2)); cnt += z.shape[0] * z.shape[1] * z.shape[2]
            zs.append(z if z.numel() * 8 * len(xs) < 40e9 else None)       # keep z when the layer fits, else recompute
        mean = s1 / cnt
        var = s2 / cnt - mean * mean
        stats[name] = (mean.cpu().numpy(), var.cpu().numpy())
        out = []
        for x, z in zip(xs, zs):
            if z is None:
                z = conv3x3_f64(pre(x), P[name + ".kernel"], P[name + ".bias"])
            out.append(post(torch.relu((z - mean) / torch.sqrt(var + 1e-3) * P[name + ".gamma"] + P[name + ".beta"])))
            del z
        return out

    def pool(x):
        n_, h, w_, c = x.shape
        return x.reshape(n_, h // 2, 2, w_ // 2, 2, c).amax(dim=(2, 4))

    def unpack(bt):   # [c][2][5000] uint32 words -> [c][400][400][2] float64
        t = torch.from_numpy(bt.view(np.int32).astype(np.int64) & 0xFFFFFFFF).to(dev)
        sh = torch.arange(32, device=dev, dtype=torch.int64)
        return ((t.unsqueeze(-1) >> sh) & 1).reshape(t.shape[0], 2, 400, 400).permute(0, 2, 3, 1).to(f64)

    xs = [bits[i:i + chunk] for i in range(0, n, chunk)]
    a = bn_block(xs, "conv1", unpack, pool)
    for i in (2, 3, 4):
        a = bn_block(a, "conv%d" % i, lambda x: x, pool)
    flat = torch.cat([x.reshape(x.shape[0], 5000) for x in a])                       # (h, w, c) = Flatten order
    f = torch.cat([torch.tensor(vec8, dtype=f64, device=dev), flat], dim=1)
    d1 = torch.relu(f @ P["dense1.kernel"] + P["dense1.bias"])
    u = torch.relu(d1 @ P["updense1.kernel"] + P["updense1.bias"]).reshape(n, 25, 25, 1)
    us = [u[i:i + chunk] for i in range(0, n, chunk)]

    def up(x):
        return upsample2(x.permute(0, 3, 1, 2), False).permute(0, 2, 3, 1)

    for j in (1, 2, 3):
        us = bn_block(us, "upconv%d" % j, up)
    return stats



if __name__ == "__main__":
    import torch
    d = np.load(sys.argv[1])
    shapes = {str(k): (int(o), tuple(int(v) for v in s if v)) for k, o, s in zip(d["names"], d["offs"], d["shp"])}
    st = batch_statistics_f64(d["w"], shapes, d["bits"], d["vec8"], device="cuda" if torch.cuda.is_available() else "cpu")
    out = {}
    for k, (m, v) in st.items():
        out[k + ".mean"], out[k + ".var"] = m, v
    np.savez(sys.argv[2], **out)
