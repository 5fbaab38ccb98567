"""Weight blob of the bi-head `pointer_model` (layout: include/ofx.h ofx_policy_layout).

No checkpoint ships with the reference (networks/ is git-ignored), so the default is the state of the model right
after construction (agents/qlearnIA_V2.py:129-186): he_uniform convolutions, glorot_uniform dense layers, zero
biases, BatchNorm gamma=1 beta=0 mean=0 var=1 - drawn from a seeded numpy RandomState."""
import numpy as np

TRUNK_CIN = (2, 8, 8, 8)
DENSE = ((5008, 100), (100, 50), (50, 2), (100, 625))
UPCONV = ((1, 2), (2, 4), (4, 8))


def layout():
    """(offsets, counts, total) - identical to ofx_policy_layout, computable without a device."""
    counts = []
    for cin in TRUNK_CIN:
        counts += [9 * cin * 8, 8, 8, 8, 8, 8]
    for fi, fo in DENSE:
        counts += [fi * fo, fo]
    for ci, co in UPCONV:
        counts += [9 * ci * co, co, co, co, co, co]
    counts += [72, 1]
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int).tolist()
    return offsets, counts, int(sum(counts))


def synthetic(seed=0x0F160002):
    off, cnt, total = layout()
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    w = np.zeros(total, np.float32)
    t = 0

    def uniform(n, lim):
        return rs.uniform(-lim, lim, n).astype(np.float32)

    for cin in TRUNK_CIN:
        w[off[t]:off[t] + cnt[t]] = uniform(cnt[t], np.sqrt(6.0 / (9 * cin)))      # he_uniform
        w[off[t + 2]:off[t + 2] + 8] = 1.0                                          # gamma
        w[off[t + 5]:off[t + 5] + 8] = 1.0                                          # moving variance
        t += 6
    for fi, fo in DENSE:
        w[off[t]:off[t] + cnt[t]] = uniform(cnt[t], np.sqrt(6.0 / (fi + fo)))       # glorot_uniform
        t += 2
    for ci, co in UPCONV:
        w[off[t]:off[t] + cnt[t]] = uniform(cnt[t], np.sqrt(6.0 / (9 * ci)))
        w[off[t + 2]:off[t + 2] + co] = 1.0
        w[off[t + 5]:off[t + 5] + co] = 1.0
        t += 6
    w[off[t]:off[t] + cnt[t]] = uniform(cnt[t], np.sqrt(6.0 / 72))
    return w
