"""How far is the HIP bi-head forward from the DECLARED graph?  Both fp32 evaluations - the C restatement
(oracle/policy_oracle.c, sequential sums) and the HIP path (BatchNorm folded, table look-up conv1, banded / phase-form
MFMA GEMMs, split-K dense1) - are measured against a float64 evaluation of the same graph (tests/policy_ref64.py).
The bounds below are 4x the largest error either of them showed on the chip (r02), per output:
act_values, heat-map interior, heat-map frame (the frame takes the zero-padding path of k_head_frames).  Parity with Keras itself
stays UNPINNED (no keras / tensorflow / weights in the image)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# max |err| / max |value| of a tensor against float64.  Measured (r02, gpurun_out/policy_fp64_report.json):
#   HIP path      act 3.6e-6 (4096 x 8, bench weights), heat interior 5.8e-7, heat frame 5.9e-7
#   C restatement act 2.0e-6,                           heat interior 2.0e-6, heat frame 1.9e-6
# arg-max equal to the float64 map's for 288 of 288 ships.
TOL_ACT, TOL_HEAT, TOL_FRAME = 1.5e-5, 8e-6, 8e-6
REPORT = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "policy_fp64_report.json")


def _rollout(N, M, seed, ticks):
    from ofighters_amd import ArenaBatch
    b = ArenaBatch(N, M)
    b.spawn_random(seed)
    for t in range(ticks):
        b.bot_actions(["turret"] * (M // 2) + ["random"] * (M - M // 2), seed, tick=t)
        b.step(actions_ptr=b._actions.ptr)
    return b


def _report(tag, rec):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        data = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        data[tag] = rec
        json.dump(data, open(REPORT, "w"), indent=1)
    except OSError:
        pass
    print(tag, rec)


@pytest.mark.parametrize("trained,legacy", [(False, False), (True, False), (True, True)])
def test_small_batch_against_fp64(trained, legacy):
    """legacy: the TF1 bilinear convention (OFX_OPT_BILINEAR_LEGACY) - other phase weights, frame lines and corner
    cells, same error bounds."""
    import torch
    from ofighters_amd import _native as nat
    from oracle import pyoracle
    from tests import policy_ref64 as R
    torch.set_num_threads(8)
    N, M = 4, 4
    b = _rollout(N, M, seed=31, ticks=40)
    w, _ = pyoracle.policy_init(5, trained_like=trained)
    b.set_option(nat.OPT_BILINEAR_LEGACY, int(legacy))
    out = b.policy_forward_host(w, want_heat=True)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    worst = np.zeros(3)
    worst_orc = np.zeros(3)
    same = 0
    for g in range(N):
        a64, h64 = R.forward(sm[g], lm[g], head[g].astype(np.float32), w, legacy_bilinear=legacy)
        for i in range(M):
            worst = np.maximum(worst, R.errors(out["act"][g, i], out["heat"][g, i], a64[i], h64[i]))
            act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w, legacy_bilinear=legacy)
            worst_orc = np.maximum(worst_orc, R.errors(act, heat, a64[i], h64[i]))
            k = int(np.argmax(h64[i]))
            gx, gy = out["ipointer"][g, i]
            same += (gx, gy) == (k % 400, k // 400)
            # where the arg-max differs from the float64 map's, it is a tie within the measured error
            assert h64[i][gy, gx] >= h64[i].max() - 2 * TOL_HEAT * float(np.abs(h64[i]).max())
            if abs(float(a64[i][0] - a64[i][1])) > 2 * TOL_ACT * max(1.0, float(np.abs(a64[i]).max())):
                assert out["iaction"][g, i] == int(np.argmax(a64[i]))
    _report(("small_trained" if trained else "small_init") + ("_legacy_bilinear" if legacy else ""),
            dict(hip=list(map(float, worst)), oracle=list(map(float, worst_orc)), argmax_same=int(same), ships=N * M))
    assert worst[0] <= TOL_ACT and worst[1] <= TOL_HEAT and worst[2] <= TOL_FRAME, worst
    assert worst_orc[0] <= TOL_ACT and worst_orc[1] <= TOL_HEAT and worst_orc[2] <= TOL_FRAME, worst_orc
    b.close()


def test_full_size_sample_against_fp64():
    """BASELINE configs[3] (4096 arenas x 8 ships) with the bench's weights synthetic(0x0F160002): 256 ships (4 in each
    of 64 random arenas) against float64 - act_values, the whole heat map, the arg-max."""
    import torch
    from ofighters_amd import DeviceBuffer, _native as nat
    from ofighters_amd.agents.policy_weights import synthetic
    from tests import policy_ref64 as R
    torch.set_num_threads(16)
    N, M = 4096, 8
    b = _rollout(N, M, seed=0x0F160001, ticks=12)
    w = synthetic(0x0F160002)
    S = N * M
    rs = np.random.RandomState(7)
    arenas = np.sort(rs.choice(N, 64, replace=False))
    mask = np.zeros((N, M), np.uint8)
    picks = {}
    for g in arenas:
        picks[int(g)] = np.sort(rs.choice(M, 4, replace=False))
        mask[g, picks[int(g)]] = 1
    dw = DeviceBuffer(w.nbytes).upload(w)
    dm = DeviceBuffer(S).upload(mask)
    da, di, dp = DeviceBuffer(8 * S), DeviceBuffer(4 * S), DeviceBuffer(8 * S)
    dh = DeviceBuffer(4 * S * 400 * 400)                 # 21 GB of the card's 288: only the masked ships are written
    b.policy_forward(dw.ptr, dm.ptr, da.ptr, di.ptr, dp.ptr, dh.ptr)
    b.sync()
    act = da.download(np.float32, (N, M, 2))
    ptr = dp.download(np.int32, (N, M, 2))
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    worst = np.zeros(3)
    same = n = 0
    for g in arenas:
        g = int(g)
        a64, h64 = R.forward(sm[g], lm[g], head[g, picks[g]].astype(np.float32), w)
        for j, i in enumerate(picks[g]):
            heat = dh.download(np.float32, (400, 400), offset=4 * (g * M + int(i)) * 160000)
            err = R.errors(act[g, i], heat, a64[j], h64[j])
            if max(err) > 1e-4:                      # diagnostics for the report
                e = np.abs(heat - h64[j])
                bad = np.argwhere(e > 1e-4 * np.abs(h64[j]).max())
                _report("bad_%d_%d" % (g, i), dict(err=list(map(float, err)), rows=[int(bad[:, 0].min()), int(bad[:, 0].max())],
                                                   cols=[int(bad[:, 1].min()), int(bad[:, 1].max())], n=int(len(bad))))
            worst = np.maximum(worst, err)
            k = int(np.argmax(h64[j]))
            gx, gy = ptr[g, i]
            same += (gx, gy) == (k % 400, k // 400)
            n += 1
            assert h64[j][gy, gx] >= h64[j].max() - 2 * TOL_HEAT * float(np.abs(h64[j]).max())
            kk = int(np.argmax(heat))                # the fused arg-max is the first maximum of the map the kernel wrote
            assert (gx, gy) == (kk % 400, kk // 400)
    _report("full_size", dict(hip=list(map(float, worst)), argmax_same=int(same), ships=int(n)))
    assert n == 256
    assert worst[0] <= TOL_ACT and worst[1] <= TOL_HEAT and worst[2] <= TOL_FRAME, worst
    dh.free()
    b.close()
