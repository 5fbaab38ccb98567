"""GPU bi-head policy forward vs the CPU restatement (oracle/policy_oracle.c).
fp32 with a different summation order (BN folded, MFMA dense, phase-form up-convolutions): tolerance 2e-5 of the
tensor's max magnitude - both sides were measured against a float64 evaluation of the graph
(tests/test_gpu_policy_fp64.py: HIP <= 3.6e-6, restatement <= 2.0e-6; 2e-5 is ~4x their sum).
Parity with Keras itself is UNPINNED (no keras/tensorflow/weights available)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _rollout(N, M, seed, ticks):
    from ofighters_amd import ArenaBatch
    b = ArenaBatch(N, M)
    b.spawn_random(seed)
    for t in range(ticks):
        b.bot_actions(["turret"] * (M // 2) + ["random"] * (M - M // 2), seed, tick=t)
        b.step(actions_ptr=b._actions.ptr)
    return b


@pytest.mark.parametrize("trained", [False, True])
def test_policy_forward_vs_oracle(trained):
    from ofighters_amd import _native as nat
    from oracle import pyoracle
    N, M = 3, 3
    b = _rollout(N, M, seed=12, ticks=35)
    w, _ = pyoracle.policy_init(3, trained_like=trained)
    off, cnt, total = b.policy_layout()
    ooff, ocnt, ototal = pyoracle.policy_layout()
    assert total == ototal == len(w) and off == list(ooff) and cnt == list(ocnt)
    out = b.policy_forward_host(w, want_heat=True)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    for g in range(N):
        for i in range(M):
            act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w)
            scale = max(1.0, float(np.abs(act).max()))
            np.testing.assert_allclose(out["act"][g, i], act, rtol=0, atol=TOL * scale)
            hs = float(np.abs(heat).max())
            np.testing.assert_allclose(out["heat"][g, i], heat, rtol=0, atol=TOL * hs)
            # frame pixels take the zero-padding correction path: check them explicitly
            for sl in (np.s_[0, :], np.s_[399, :], np.s_[:, 0], np.s_[:, 399]):
                np.testing.assert_allclose(out["heat"][g, i][sl], heat[sl], rtol=0, atol=TOL * hs)
            if abs(float(act[0] - act[1])) > 2 * TOL * scale:
                assert out["iaction"][g, i] == ia
            gx, gy = out["ipointer"][g, i]
            # the GPU's arg-max is the first maximum of ITS heat-map ...
            k = int(np.argmax(out["heat"][g, i]))
            assert (gx, gy) == (k % 400, k // 400)
            # ... and a maximum of the oracle's up to the tolerance
            assert heat[gy, gx] >= heat.max() - 2 * TOL * hs
    b.close()


def test_policy_mask_and_actions():
    """ship_mask selects the policy ships (reference default: one QlearnIA ship per
    arena, lib/ofighters.py:53); QlearnIA.play packing: exactly one of shoot/thrust."""
    from ofighters_amd import DeviceBuffer, _native as nat
    from oracle import pyoracle
    N, M = 4, 4
    b = _rollout(N, M, seed=5, ticks=20)
    w, _ = pyoracle.policy_init(9, trained_like=True)
    mask = np.zeros((N, M), np.uint8)
    mask[:, 1] = 1
    full = b.policy_forward_host(w)
    part = b.policy_forward_host(w, ship_mask=mask)
    assert np.array_equal(part["ipointer"][:, 1], full["ipointer"][:, 1])
    assert np.array_equal(part["iaction"][:, 1], full["iaction"][:, 1])
    np.testing.assert_array_equal(part["act"][:, 1], full["act"][:, 1])
    # action packing from the workspace results of a full forward
    dw = DeviceBuffer(w.nbytes).upload(w)
    b.policy_forward(dw.ptr)
    b.policy_actions()
    acts = b.actions_host()
    alive = b.get(nat.F_SHIP_ALIVE)
    assert np.array_equal(acts["valid"], alive)
    assert np.all(acts["shoot"] + acts["thrust"] == 1)
    assert np.array_equal(acts["shoot"], (full["iaction"] == 0).astype(np.uint8))
    assert np.array_equal(acts["px"], full["ipointer"][..., 0]) and np.array_equal(acts["py"], full["ipointer"][..., 1])
    b.step(actions_ptr=b._actions.ptr)      # the packed actions drive a tick
    b.close()


def test_policy_determinism_and_trunk_sharing():
    """Two runs give identical bits; ships of one arena differ only through their
    8-scalar head (the trunk is computed once per arena)."""
    from oracle import pyoracle
    b = _rollout(2, 8, seed=3, ticks=25)
    w, _ = pyoracle.policy_init(4, trained_like=True)
    a1 = b.policy_forward_host(w)
    a2 = b.policy_forward_host(w)
    for k in a1:
        assert np.array_equal(a1[k], a2[k])
    b.close()


def test_reference_variants_agree():
    """Two plain variants are kept as references (ofx_set_option): OFX_OPT_TRUNK_PLAIN runs the four trunk layers
    through the VALU convolution instead of the table look-up / banded MFMA kernels, OFX_OPT_FRAMES_REF computes the
    frame lines of the head from the definition (up-sample, zero-padded convolution) instead of the phase form.  They
    agree with the production path up to fp32 summation order."""
    from ofighters_amd import _native as nat
    from oracle import pyoracle
    b = _rollout(4, 8, seed=8, ticks=30)
    w, _ = pyoracle.policy_init(6, trained_like=True)
    base = b.policy_forward_host(w, want_heat=True)
    hs = float(np.abs(base["heat"]).max())
    for opt in (nat.OPT_TRUNK_PLAIN, nat.OPT_FRAMES_REF):
        b.set_option(opt, 1)
        other = b.policy_forward_host(w, want_heat=True)
        b.set_option(opt, 0)
        np.testing.assert_allclose(other["heat"], base["heat"], rtol=0, atol=TOL * hs)
        np.testing.assert_allclose(other["act"], base["act"], rtol=0, atol=TOL * max(1.0, float(np.abs(base["act"]).max())))
        if opt == nat.OPT_FRAMES_REF:   # only the frame cells of the head change: the interior is bit-identical
            assert np.array_equal(other["act"], base["act"])
    again = b.policy_forward_host(w, want_heat=True)
    for k in base:
        assert np.array_equal(again[k], base[k]), k
    b.close()


def test_trunk_fused_matches_split():
    """OFX_OPT_TRUNK_FUSE: conv1 -> conv2 in one persistent kernel (k_trunk12: the pooled conv1 activation lives only in
    LDS) against the two-kernel form - same table sums, same banded-GEMM k order, so the results are bit-identical.
    300 arenas on 256 CUs: some workgroups walk two images (the persistent loop), some one."""
    from ofighters_amd import _native as nat
    from oracle import pyoracle
    N, M = 300, 2
    b = _rollout(N, M, seed=19, ticks=25)
    w, _ = pyoracle.policy_init(7, trained_like=True)
    b.set_option(nat.OPT_TRUNK_FUSE, 2)
    split = b.policy_forward_host(w)
    b.set_option(nat.OPT_TRUNK_FUSE, 1)
    fused = b.policy_forward_host(w)
    b.set_option(nat.OPT_TRUNK_FUSE, 0)
    for k in split:
        assert np.array_equal(fused[k], split[k]), k
    # fewer images than CUs
    for n_small in (1, 2, 5):
        bs = _rollout(n_small, M, seed=23 + n_small, ticks=10)
        bs.set_option(nat.OPT_TRUNK_FUSE, 2)
        ref = bs.policy_forward_host(w)
        bs.set_option(nat.OPT_TRUNK_FUSE, 1)
        got = bs.policy_forward_host(w)
        for k in ref:
            assert np.array_equal(got[k], ref[k]), (n_small, k)
        bs.close()
    # and against the restatement for a few ships (the fused path is the one the full-size workloads run)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    for g, i in ((0, 0), (255, 1), (256, 0), (299, 1)):
        act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w)
        np.testing.assert_allclose(fused["act"][g, i], act, rtol=0, atol=TOL * max(1.0, float(np.abs(act).max())))
        gx, gy = fused["ipointer"][g, i]
        assert heat[gy, gx] >= heat.max() - 2 * TOL * float(np.abs(heat).max())
    b.close()


def test_trunk_sparse_is_bit_identical():
    """OFX_OPT_TRUNK_SPARSE (opt-in, exact): the streaming trunk writes conv1's constant for waves of empty bit windows and
    stores conv2's constant for M-tiles whose whole window holds it - the dense kernel's own bits, so every output of the
    forward is EQUAL (==), on a mid-episode rollout (300 arenas: the persistent loop), on fresh spawns (no lasers), with
    ships pushed against every border and corner, on a brawl packed into one corner, and for 1 / 2 / 5 images; the heat map
    too.  The counters show that work was actually skipped."""
    from ofighters_amd import ArenaBatch, _native as nat
    from oracle import pyoracle
    w, _ = pyoracle.policy_init(7, trained_like=True)

    def both(b, want_heat=False):
        b.set_option(nat.OPT_TRUNK_FUSE, 1)
        b.set_option(nat.OPT_TRUNK_SPARSE, 0)
        dense = b.policy_forward_host(w, want_heat=want_heat)
        b.set_option(nat.OPT_TRUNK_SPARSE, 1)
        b.policy_trunk_stats()
        sparse = b.policy_forward_host(w, want_heat=want_heat)
        st = b.policy_trunk_stats()
        b.set_option(nat.OPT_TRUNK_SPARSE, 0)
        for k in dense:
            assert np.array_equal(sparse[k], dense[k]), k
        return st

    N, M = 300, 2
    b = _rollout(N, M, seed=19, ticks=25)
    run, total, trun, ttotal = both(b)
    assert total == N * 20 * 63 and 0 < run < 0.7 * total, (run, total)          # 63 M-tiles per step, 20 steps per image
    assert ttotal > 0 and 0 < trun < 0.8 * ttotal, (trun, ttotal)
    # ships on the borders and in the corners of the map (x, y in {0, 1, 199 .. 400}), lasers flying
    rs = np.random.RandomState(5)
    edge = np.array([0, 1, 2, 7, 8, 199, 200, 391, 392, 398, 399, 400])
    b.set_ships(x=rs.choice(edge, (N, M)), y=rs.choice(edge, (N, M)))
    both(b)
    b.close()
    fresh = ArenaBatch(64, 8)
    fresh.spawn_random(3)                                                        # tick 0: ships only
    run, total, _, _ = both(fresh)
    assert run < 0.5 * total
    # a brawl in one corner: everything non-constant there, nothing elsewhere
    fresh.set_ships(x=rs.randint(0, 40, (64, 8)), y=rs.randint(360, 401, (64, 8)))
    for t in range(12):
        fresh.bot_actions(["turret"] * 8, 5, tick=t)
        fresh.step(actions_ptr=fresh._actions.ptr)
    both(fresh)
    fresh.close()
    for n_small in (1, 2, 5):
        bs = _rollout(n_small, M, seed=23 + n_small, ticks=30)
        both(bs, want_heat=True)
        bs.close()
    # with 16-bit operands the constant is the 16-bit sequence's own: still equal to that mode's dense form
    lp = _rollout(40, M, seed=31, ticks=30)
    for mode in (1, 2):
        lp.set_option(nat.OPT_POLICY_BF16, mode)
        both(lp)
    lp.close()


def test_legacy_bilinear_option():
    """OFX_OPT_BILINEAR_LEGACY: the TF1 resize_bilinear convention of UpSampling2D (src = dst / 2) - the second
    meaning the reference's unpinned keras range admits (qlearnIA_V2.py:166-184).  Against the restatement with the
    same switch, frame pixels included, through both frame-line kernels, pinned and unpinned; and it IS a different
    function from the default."""
    from ofighters_amd import DeviceBuffer, _native as nat
    from oracle import pyoracle
    N, M = 2, 3
    b = _rollout(N, M, seed=17, ticks=30)
    w, _ = pyoracle.policy_init(8, trained_like=True)
    default = b.policy_forward_host(w, want_heat=True)
    b.set_option(nat.OPT_BILINEAR_LEGACY, 1)
    out = b.policy_forward_host(w, want_heat=True)
    b.set_option(nat.OPT_FRAMES_REF, 1)
    out_ref = b.policy_forward_host(w, want_heat=True)
    b.set_option(nat.OPT_FRAMES_REF, 0)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    for g in range(N):
        for i in range(M):
            act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w, legacy_bilinear=True)
            hs = float(np.abs(heat).max())
            for o in (out, out_ref):
                np.testing.assert_allclose(o["act"][g, i], act, rtol=0, atol=TOL * max(1.0, float(np.abs(act).max())))
                np.testing.assert_allclose(o["heat"][g, i], heat, rtol=0, atol=TOL * hs)
                gx, gy = o["ipointer"][g, i]
                assert heat[gy, gx] >= heat.max() - 2 * TOL * hs
            assert np.abs(default["heat"][g, i] - heat).max() > 1e-3 * hs
    # a pinned blob is re-prepared when the convention changes
    S = N * M
    dw = DeviceBuffer(w.nbytes).upload(w)
    dp = DeviceBuffer(8 * S)
    b.policy_pin_weights(dw.ptr)
    b.policy_forward(dw.ptr, None, None, None, dp.ptr, None); b.sync()
    assert np.array_equal(dp.download(np.int32, (N, M, 2)), out["ipointer"])
    b.set_option(nat.OPT_BILINEAR_LEGACY, 0)
    b.policy_forward(dw.ptr, None, None, None, dp.ptr, None); b.sync()
    assert np.array_equal(dp.download(np.int32, (N, M, 2)), default["ipointer"])
    b.close()


def test_pinned_weights():
    """ofx_policy_pin_weights: the prepared weights (BatchNorm folded, phase weights, tables) are built once and reused
    by every forward on the same blob; a pinned forward is bit-identical to an unpinned one, and pinning again picks
    up new values written into the blob."""
    from ofighters_amd import DeviceBuffer
    from oracle import pyoracle
    N, M = 3, 4
    b = _rollout(N, M, seed=21, ticks=15)
    w1, _ = pyoracle.policy_init(11, trained_like=True)
    w2, _ = pyoracle.policy_init(12, trained_like=True)
    S = N * M
    dw = DeviceBuffer(w1.nbytes).upload(w1)
    da, dp = DeviceBuffer(8 * S), DeviceBuffer(8 * S)

    def run():
        b.policy_forward(dw.ptr, None, da.ptr, None, dp.ptr, None)
        b.sync()
        return da.download(np.float32, (N, M, 2)), dp.download(np.int32, (N, M, 2))

    a0, p0 = run()                       # unpinned: prepared on the fly
    b.policy_pin_weights(dw.ptr)
    a1, p1 = run()
    a2, p2 = run()
    assert np.array_equal(a0, a1) and np.array_equal(p0, p1) and np.array_equal(a1, a2) and np.array_equal(p1, p2)
    dw.upload(w2)                        # new values in the pinned blob: the dense layers read the blob directly, the
    b.policy_pin_weights(dw.ptr)         # convolutions their prepared copy - pin again
    a3, p3 = run()
    b.policy_pin_weights(None)
    a4, p4 = run()
    assert np.array_equal(a3, a4) and np.array_equal(p3, p4)
    ref = b.policy_forward_host(w2)
    assert np.array_equal(ref["act"], a4) and np.array_equal(ref["ipointer"], p4)
    assert not np.array_equal(a0, a4)
    b.close()


def test_pinned_blob_survives_a_forward_on_another_blob():
    """A target network next to a pinned online network: with blob A pinned, a forward that names blob B must not
    overwrite A's prepared weights - the next forward on A is bit-identical to an unpinned forward on A.  Also the
    heat map's values: a NaN weight gives np.argmax's answer (0, 0) for an all-NaN map, never an out-of-range pointer."""
    from ofighters_amd import DeviceBuffer
    from oracle import pyoracle
    N, M = 3, 4
    b = _rollout(N, M, seed=22, ticks=15)
    wa, _ = pyoracle.policy_init(13, trained_like=True)
    wb, _ = pyoracle.policy_init(14, trained_like=True)
    S = N * M
    da_, db_ = DeviceBuffer(wa.nbytes).upload(wa), DeviceBuffer(wb.nbytes).upload(wb)
    oa, op = DeviceBuffer(8 * S), DeviceBuffer(8 * S)

    def run(w):
        b.policy_forward(w.ptr, None, oa.ptr, None, op.ptr, None)
        b.sync()
        return oa.download(np.float32, (N, M, 2)), op.download(np.int32, (N, M, 2))

    a_ref, p_ref = run(da_)              # unpinned
    b_ref, q_ref = run(db_)
    b.policy_pin_weights(da_.ptr)
    a1, p1 = run(da_)
    b2, q2 = run(db_)                    # another blob while A is pinned
    a3, p3 = run(da_)                    # A again: must still be A's preparation
    assert np.array_equal(a_ref, a1) and np.array_equal(p_ref, p1)
    assert np.array_equal(b_ref, b2) and np.array_equal(q_ref, q2)
    assert np.array_equal(a_ref, a3) and np.array_equal(p_ref, p3)
    assert not np.array_equal(p_ref, q_ref)
    # NaN in the last convolution's bias: every heat-map value is NaN
    off, cnt, _ = b.policy_layout()
    wn = wa.copy()
    wn[off[51]] = np.nan
    dn = DeviceBuffer(wn.nbytes).upload(wn)
    _, pn = run(dn)
    assert np.array_equal(pn, np.zeros_like(pn))
    b.close()


def test_policy_full_size_properties():
    """BASELINE configs[3] size (4096 x 8): determinism, mask consistency (a masked forward equals the same ships of
    a full forward), pointer range, and a sample of ships against the CPU restatement."""
    from ofighters_amd import DeviceBuffer, _native as nat
    from ofighters_amd.agents.policy_weights import synthetic
    from oracle import pyoracle
    N, M = 4096, 8
    b = _rollout(N, M, seed=0x0F160001, ticks=12)
    w = synthetic()
    S = N * M
    dw = DeviceBuffer(w.nbytes).upload(w)
    da, di, dp = DeviceBuffer(8 * S), DeviceBuffer(4 * S), DeviceBuffer(8 * S)

    def run(mask_ptr=None):
        b.policy_forward(dw.ptr, mask_ptr, da.ptr, di.ptr, dp.ptr, None)
        b.sync()
        return (da.download(np.float32, (N, M, 2)), di.download(np.int32, (N, M)), dp.download(np.int32, (N, M, 2)))

    a1, i1, p1 = run()
    a2, i2, p2 = run()
    assert np.array_equal(a1, a2) and np.array_equal(i1, i2) and np.array_equal(p1, p2)     # bitwise reproducible
    assert p1.min() >= 0 and p1.max() <= 399 and set(np.unique(i1)) <= {0, 1}
    mask = np.zeros((N, M), np.uint8)
    mask[:, 3] = 1
    dm = DeviceBuffer(S).upload(mask)
    a3, i3, p3 = run(dm.ptr)
    assert np.array_equal(a3[:, 3], a1[:, 3]) and np.array_equal(p3[:, 3], p1[:, 3])
    # the streaming trunk (the default at this size) against the split kernels: bit-identical for all 32768 ships
    b.set_option(nat.OPT_TRUNK_FUSE, 2)
    a4, i4, p4 = run()
    b.set_option(nat.OPT_TRUNK_FUSE, 0)
    assert np.array_equal(a4, a1) and np.array_equal(i4, i1) and np.array_equal(p4, p1)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    rs = np.random.RandomState(2)
    for g in rs.choice(N, 3, replace=False):
        i = int(rs.randint(M))
        act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w)
        np.testing.assert_allclose(a1[g, i], act, rtol=0, atol=TOL * max(1.0, float(np.abs(act).max())))
        gx, gy = p1[g, i]
        assert heat[gy, gx] >= heat.max() - 2 * TOL * float(np.abs(heat).max())
    b.close()
