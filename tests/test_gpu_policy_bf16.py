"""OFX_OPT_POLICY_BF16 - the OPT-IN reduced-precision forward (value 1: bf16, value 2: fp16 operands in conv2-4 and
upconv3 / upconv4, fp32 sums; never the default, never the headline number).  Measured against the float64 graph like the fp32 path (tests/policy_ref64.py):
the error of the heat map and how often its arg-max is the float64 map's are REPORTED (gpurun_out/policy_fp64_report.json,
bench.py repeats the measurement in its bf16 lines); asserted is only that the switch gives a reduced-precision version
of the same function - act_values within 2e-2 of their scale (conv2 / conv3 of the streaming trunk run on bf16 operands
too), the heat map within 3e-2 of its scale, the fused arg-max equal to the arg-max of the map the kernel wrote, every
pointer a near-maximum of the float64 map - and that the fp32 path is untouched by the switch having been used."""
import numpy as np
import pytest

from tests.test_gpu_policy_fp64 import _report, _rollout

pytestmark = pytest.mark.gpu
TOL_LOWP = {1: 3e-2, 2: 4e-3}   # max |heat - heat64| / max |heat64|; measured 8e-3 (bf16: 8 significant bits) / ~1e-3 (fp16: 11)


@pytest.mark.parametrize("lowp", [1, 2])
def test_bf16_forward_against_fp64_and_fp32(lowp):
    import torch
    from ofighters_amd import _native as nat
    from oracle import pyoracle
    from tests import policy_ref64 as R
    torch.set_num_threads(8)
    N, M = 4, 4
    b = _rollout(N, M, seed=31, ticks=40)
    w, _ = pyoracle.policy_init(5, trained_like=True)
    b.set_option(nat.OPT_TRUNK_FUSE, 1)      # the streaming trunk (what large batches run): its conv2 / conv3 take the switch too
    fp32 = b.policy_forward_host(w, want_heat=True)
    TOL_BF16 = TOL_LOWP[lowp]
    b.set_option(nat.OPT_POLICY_BF16, lowp)
    bf = b.policy_forward_host(w, want_heat=True)
    mask = np.zeros((N, M), np.uint8)
    mask[:, 1] = 1
    bf_masked = b.policy_forward_host(w, ship_mask=mask)            # the instantiation without the heat-map output
    b.set_option(nat.OPT_POLICY_BF16, 0)
    again = b.policy_forward_host(w, want_heat=True)
    for k in fp32:
        assert np.array_equal(fp32[k], again[k]), k                  # the fp32 path is what it was
    act_scale = float(np.abs(fp32["act"]).max())
    act_err = float(np.abs(bf["act"] - fp32["act"]).max()) / act_scale
    assert 0 < act_err <= (2e-2 if lowp == 1 else 3e-3), act_err
    assert np.array_equal(bf_masked["ipointer"][:, 1], bf["ipointer"][:, 1])
    assert not np.array_equal(bf["heat"], fp32["heat"])
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    worst, same, same32, rel32 = 0.0, 0, 0, 0.0
    for g in range(N):
        a64, h64 = R.forward(sm[g], lm[g], head[g].astype(np.float32), w)
        for i in range(M):
            scale = float(np.abs(h64[i]).max())
            err = float(np.abs(bf["heat"][g, i] - h64[i]).max()) / scale
            worst = max(worst, err)
            rel32 = max(rel32, float(np.abs(bf["heat"][g, i] - fp32["heat"][g, i]).max()) / scale)
            k = int(np.argmax(h64[i]))
            gx, gy = bf["ipointer"][g, i]
            same += (gx, gy) == (k % 400, k // 400)
            same32 += tuple(bf["ipointer"][g, i]) == tuple(fp32["ipointer"][g, i])
            kk = int(np.argmax(bf["heat"][g, i]))
            assert (gx, gy) == (kk % 400, kk // 400)                  # first maximum of the map the kernel wrote
            assert h64[i][gy, gx] >= h64[i].max() - 2 * TOL_BF16 * scale
    _report("small_trained_" + ("bf16" if lowp == 1 else "fp16"), dict(heat_err=worst, heat_vs_fp32=rel32, act_vs_fp32=act_err, argmax_same_as_fp64=int(same),
                                       argmax_same_as_fp32=int(same32), ships=N * M))
    assert 1e-5 < worst <= TOL_BF16, worst
    with pytest.raises(Exception):
        b.set_option(nat.OPT_POLICY_BF16, 3)
    b.close()
