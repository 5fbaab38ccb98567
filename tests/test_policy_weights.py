"""Keras interchange of the policy blob (tensor order only; element order is Keras' own)."""
import numpy as np
import pytest

from ofighters_amd.agents import policy_weights as pw


def _random_blob(seed):
    off, cnt, total = pw.layout()
    return np.random.RandomState(seed).standard_normal(total).astype(np.float32)


def test_roundtrip_creation_order(tmp_path):
    blob = _random_blob(1)
    ws = pw.to_keras(blob)
    assert ws[0].shape == (3, 3, 2, 8) and ws[24].shape == (5008, 100) and ws[-2].shape == (3, 3, 8, 1)
    assert np.array_equal(pw.from_keras(ws), blob)
    pw.save_npz(tmp_path / "w.npz", blob)
    assert np.array_equal(pw.load_npz(tmp_path / "w.npz"), blob)


def test_depth_sorted_order():
    """Keras sorts functional-model layers by depth: the head-1 dense layers interleave with head 2's."""
    blob = _random_blob(2)
    ws = pw.to_keras(blob)
    trunk, dense1 = ws[:24], ws[24:26]
    dense2, out1, updense = ws[26:28], ws[28:30], ws[30:32]
    up = ws[32:]
    order = trunk + dense1 + updense + dense2 + up[:2] + out1 + up[2:]
    assert np.array_equal(pw.from_keras(order), blob)


def test_rejects_bad_lists():
    ws = pw.to_keras(_random_blob(3))
    with pytest.raises(Exception, match="missing tensors"):
        pw.from_keras(ws[:-2])
    with pytest.raises(Exception):
        pw.from_keras(ws + [np.zeros((3, 3, 8, 8), np.float32), np.zeros(8, np.float32)])
    bad = list(ws)
    bad[0] = np.zeros((3, 3, 3, 8), np.float32)
    with pytest.raises(Exception, match="no free slot"):
        pw.from_keras(bad)


def test_trainer_save_load(tmp_path):
    from ofighters_amd.agents.qlearn import Trainer
    t = Trainer(weights=_random_blob(4))
    path = t.save(id="iteration-3", folder=str(tmp_path), name="t")
    assert path.endswith("keras-model-t-iteration-3.npz")
    with pytest.raises(Exception, match="exists"):
        t.save(id="iteration-3", folder=str(tmp_path), name="t")
    u = Trainer()
    u.load(path)
    assert np.array_equal(u.weights, t.weights)
