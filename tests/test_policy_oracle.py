"""Cross-check of the bi-head policy restatement (oracle/policy_oracle.c) against
torch CPU ops.  This is NOT reference parity (keras/tensorflow are absent and no
weights ship: parity of P1 is unpinned) - it checks that the
C restatement computes the declared graph."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pyoracle


def torch_forward(sm, lm, vec8, w, lay, legacy=False):
    from tests.policy_ref64 import upsample2
    def T(name):
        o, shp = lay[name]
        return torch.from_numpy(w[o:o + int(np.prod(shp))].reshape(shp).copy())

    def conv(x, name):                              # HWIO -> OIHW
        return F.conv2d(x, T(name + ".kernel").permute(3, 2, 0, 1), T(name + ".bias"), padding=1)

    def bn(x, name):
        return F.batch_norm(x, T(name + ".mean"), T(name + ".var"), T(name + ".gamma"), T(name + ".beta"),
                            training=False, eps=1e-3)

    x = torch.from_numpy(np.stack([sm, lm], 0).astype(np.float32))[None]      # NCHW
    for i in (1, 2, 3, 4):
        x = F.max_pool2d(F.relu(bn(conv(x, "conv%d" % i), "conv%d" % i)), 2)
    flat = x.permute(0, 2, 3, 1).reshape(1, -1)                              # Flatten (h,w,c)
    cat = torch.cat([torch.from_numpy(vec8.astype(np.float32))[None], flat], 1)
    d1 = F.relu(cat @ T("dense1.kernel") + T("dense1.bias"))
    d2 = F.relu(d1 @ T("dense2.kernel") + T("dense2.bias"))
    act = d2 @ T("output1.kernel") + T("output1.bias")
    u = F.relu(d1 @ T("updense1.kernel") + T("updense1.bias")).reshape(1, 1, 25, 25)
    for i in (1, 2, 3):
        u = upsample2(u, legacy)
        u = F.relu(bn(conv(u, "upconv%d" % i), "upconv%d" % i))
    u = upsample2(u, legacy)
    heat = conv(u, "upconv4")[0, 0]
    return act[0].numpy(), heat.numpy()


def scene(seed):
    rs = np.random.RandomState(seed)
    sm = np.zeros((400, 400), np.uint8)
    lm = np.zeros((400, 400), np.uint8)
    for _ in range(8):
        x, y = rs.randint(0, 401, 2)
        sm |= pyoracle.disk(float(y), float(x), 8.0)
    for _ in range(30):
        x, y = rs.uniform(0, 400, 2)
        lm |= pyoracle.disk(y, x, 2.0)
    vec8 = np.array([rs.randint(0, 4), 1, rs.randint(0, 401), rs.randint(0, 401), 400, 400,
                     rs.randint(0, 400), rs.randint(0, 400)], np.float32)
    return sm, lm, vec8


def test_layout():
    off, cnt, total = pyoracle.policy_layout()
    assert len(off) == 52 and len(pyoracle.POLICY_TENSORS) == 52
    # conv 152+584*3, dense1 500900, dense2 5050, out1 102, updense 63125, upconvs 20+76+296+73
    assert total == (152 + 3 * 584 + 4 * 32) + 500900 + 5050 + 102 + 63125 + (20 + 76 + 296 + 73) + (2 + 4 + 8) * 4
    assert total == 571730                      # ~571 k params


def test_oracle_matches_torch():
    """Both bilinear conventions the reference's version range admits (policy_oracle.c header): half-pixel centres
    against torch's interpolate, the TF1 legacy mapping against an explicit torch gather."""
    torch.set_num_threads(4)
    for seed, trained, legacy in ((0, False, False), (1, True, False), (1, True, True)):
        w, lay = pyoracle.policy_init(seed, trained_like=trained)
        sm, lm, vec8 = scene(seed)
        act, heat, ia, ip = pyoracle.policy_forward(sm, lm, vec8, w, legacy_bilinear=legacy)
        tact, theat = torch_forward(sm, lm, vec8, w, lay, legacy)
        np.testing.assert_allclose(act, tact, rtol=0, atol=2e-5 * max(1.0, float(np.abs(tact).max())))
        np.testing.assert_allclose(heat, theat, rtol=0, atol=2e-5 * float(np.abs(theat).max()))
        assert ia == int(np.argmax(act))
        k = int(np.argmax(heat))
        assert ip == (k % 400, k // 400)                     # (x, y): unravel_index(order='F')
        assert ip == tuple(int(v) for v in np.unravel_index(k, (400, 400), order="F"))
        # the torch heat-map agrees on the arg-max up to fp32 noise
        assert theat[ip[1], ip[0]] >= theat.max() - 2e-5 * float(np.abs(theat).max())
        if legacy:   # the two conventions are different functions, not rounding variants of one
            other = pyoracle.policy_forward(sm, lm, vec8, w)[1]
            assert np.abs(other - heat).max() > 1e-3 * float(np.abs(heat).max())


def test_legacy_upsample_definition():
    """The legacy mapping on a ramp: even outputs copy, odd outputs average with the next sample, the last one clamps
    (resize_bilinear without half-pixel centres: src = dst * in/out, lower = floor, upper = min(lower + 1, n - 1))."""
    from tests.policy_ref64 import upsample2
    u = torch.arange(5, dtype=torch.float64).reshape(1, 1, 1, 5).expand(1, 1, 5, 5).contiguous()
    o = upsample2(u, True)[0, 0, 0].numpy()
    assert o.tolist() == [0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4]
    h = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=False)[0, 0, 0].numpy()
    assert h.tolist() == [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.25, 3.75, 4]


def test_oracle_error_against_float64():
    """The restatement's own fp32 error against a float64 evaluation of the declared graph (tests/policy_ref64.py):
    max |err| / max |value| <= 1.5e-5 (act), 8e-6 (heat map interior and frame) - the same bounds the HIP path is
    held to in tests/test_gpu_policy_fp64.py (measured: 2.0e-6 / 2.0e-6 / 1.9e-6; bench weights 5.3e-6 on act)."""
    from ofighters_amd.agents.policy_weights import synthetic
    from tests import policy_ref64 as R
    torch.set_num_threads(4)
    for w, legacy in ((pyoracle.policy_init(3)[0], False), (pyoracle.policy_init(3, trained_like=True)[0], False),
                      (synthetic(0x0F160002), False), (synthetic(0x0F160002), True)):
        for seed in range(2):
            sm, lm, vec8 = scene(seed)
            act, heat, ia, ip = pyoracle.policy_forward(sm, lm, vec8, w, legacy_bilinear=legacy)
            a64, h64 = R.forward(sm, lm, vec8[None], w, legacy_bilinear=legacy)
            e = R.errors(act, heat, a64[0], h64[0])
            assert e[0] <= 1.5e-5 and e[1] <= 8e-6 and e[2] <= 8e-6, e
            k = int(np.argmax(h64[0]))
            assert h64[0][ip[1], ip[0]] >= h64[0].max() - 1.6e-5 * float(np.abs(h64[0]).max())
