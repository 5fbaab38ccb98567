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


# ---- Keras interchange ("Keras-weight import") --------------------------------------------------
# The blob keeps every tensor in Keras' own element order (HWIO kernels, (in, out) dense kernels, BatchNorm as
# gamma, beta, moving_mean, moving_variance), so import / export is a matter of tensor ORDER only.  h5py is not
# part of this image: a user exports `model.get_weights()` of the reference's Trainer.model
# (agents/qlearnIA_V2.py:123-190) in their Keras environment, e.g. `np.savez("w.npz", *model.get_weights())`.

def _tensor_shapes():
    shapes = []
    for cin in TRUNK_CIN:
        shapes += [(3, 3, cin, 8), (8,), (8,), (8,), (8,), (8,)]
    for fi, fo in DENSE:
        shapes += [(fi, fo), (fo,)]
    for ci, co in UPCONV:
        shapes += [(3, 3, ci, co), (co,), (co,), (co,), (co,), (co,)]
    shapes += [(3, 3, 8, 1), (1,)]
    return shapes


def to_keras(blob):
    """Blob -> list of arrays in the model's layer-creation order (conv, BN, ... as written in
    agents/qlearnIA_V2.py:129-186), i.e. what `model.set_weights` takes when `model.layers` is in that order."""
    off, cnt, total = layout()
    blob = np.asarray(blob, np.float32)
    if blob.shape != (total,):
        raise Exception("policy blob must have %d floats" % total)
    return [blob[o:o + c].reshape(s).copy() for o, c, s in zip(off, cnt, _tensor_shapes())]


def from_keras(weights):
    """List of arrays as returned by `model.get_weights()` -> blob.

    Keras orders a functional model's layers by depth, so the layers of the two heads may interleave; the arrays of
    one layer always stay together (kernel, bias / gamma, beta, mean, variance).  Tensors are therefore matched by
    rank and shape: a 4-D array is the next convolution of its shape (the three (3,3,8,8) trunk kernels in order of
    appearance), a 2-D array the dense layer of that shape, the 1-D array behind a kernel its bias, and a run of four
    1-D arrays the BatchNorm of the earliest convolution that has none yet.  Anything else raises."""
    off, cnt, total = layout()
    shapes = _tensor_shapes()
    blob = np.zeros(total, np.float32)
    filled = [False] * len(shapes)

    def put(t, a):
        a = np.asarray(a, np.float32)
        if a.shape != shapes[t] or filled[t]:
            raise Exception("unexpected tensor of shape %s for slot %d %s" % (a.shape, t, shapes[t]))
        blob[off[t]:off[t] + cnt[t]] = a.ravel()
        filled[t] = True

    convs = [t for t, s in enumerate(shapes) if len(s) == 4]        # kernel slots in creation order
    bn_of = [t for t in convs if shapes[t] != (3, 3, 8, 1)]          # the last convolution has no BatchNorm
    bn_next = 0
    ws = [np.asarray(a) for a in weights]
    i = 0
    while i < len(ws):
        a = ws[i]
        if a.ndim in (4, 2):
            cand = [t for t, s in enumerate(shapes) if s == tuple(a.shape) and not filled[t]]
            if not cand or i + 1 >= len(ws):
                raise Exception("no free slot for a kernel of shape %s" % (a.shape,))
            put(cand[0], a)
            put(cand[0] + 1, ws[i + 1])
            i += 2
        elif a.ndim == 1:
            if bn_next >= len(bn_of) or i + 4 > len(ws):
                raise Exception("unexpected vector of length %d at position %d" % (a.shape[0], i))
            t = bn_of[bn_next]
            if not filled[t]:
                raise Exception("BatchNorm statistics before their convolution (position %d)" % i)
            for k in range(4):
                put(t + 2 + k, ws[i + k])
            bn_next += 1
            i += 4
        else:
            raise Exception("unexpected tensor rank %d" % a.ndim)
    if not all(filled):
        raise Exception("missing tensors: slots %s" % [t for t, f in enumerate(filled) if not f])
    return blob


def load_npz(path):
    """`np.savez(path, *model.get_weights())` written in the user's Keras environment -> blob."""
    with np.load(path, allow_pickle=False) as z:
        return from_keras([z["arr_%d" % i] for i in range(len(z.files))])


def save_npz(path, blob):
    np.savez(path, *to_keras(blob))
