"""(f) rank 2: exploration schedules, epsilon-greedy / collecting random play, and the QlearnIA agent flow."""
import random

import numpy as np
import pytest

from ofighters_amd.lib.epsilon import Epsilon_cos, Epsilon_decay
from tests.trace_util import GOLDEN


def _golden():
    zf = np.load(GOLDEN + "/epsilon.npz")
    return {k: zf[k] for k in zf.files}


def test_epsilon_schedules_match_reference():
    z = _golden()
    e = Epsilon_cos(period=110 * 400)
    assert np.array_equal(np.array([e.get()] + [e.next() for _ in range(600)]), z["cos_44000_first"])
    e = Epsilon_cos(period=50)
    assert np.array_equal(np.array([e.get()] + [e.next() for _ in range(130)]), z["cos_50"])
    e.set(0.25)
    assert np.array_equal(np.array([e.t, e.get()] + [e.next() for _ in range(5)]), z["cos_50_after_set"])
    d = Epsilon_decay()
    assert np.array_equal(np.array([d.get()] + [d.next() for _ in range(3000)]), z["decay"])
    d.set(0.0105)
    assert np.array_equal(np.array([d.next() for _ in range(800)]), z["decay_after_set"])
    with pytest.raises(Exception, match="range"):
        d.set(1.5)


def test_policy_layout_host_equals_oracle():
    from ofighters_amd.agents import policy_weights
    from oracle import pyoracle
    off, cnt, total = policy_weights.layout()
    ooff, ocnt, ototal = pyoracle.policy_layout()
    assert off == list(ooff) and cnt == list(ocnt) and total == ototal
    w = policy_weights.synthetic()
    assert w.dtype == np.float32 and w.shape == (total,) and np.isfinite(w).all()


def test_qlearnia_flow_on_cpu_engine():
    """The reference's default line-up {"idle": 6, "QlearnIA": 1} (lib/ofighters.py:53) through the facade:
    19 collecting random plays, then epsilon-greedy forwards; exactly one of shoot/thrust per action."""
    from ofighters_amd.lib.battleground import Battleground
    from ofighters_amd.lib.epsilon import Epsilon_decay
    from ofighters_amd.agents import qlearn
    from tests.oracle_batch import OracleEngine
    random.seed(4)
    np.random.seed(4)
    qlearn.TRAINER = qlearn.Trainer(epsilon=Epsilon_decay())
    qlearn.TRAINER.epsilon.set(0.5)
    bg = Battleground(ships={"idle": 2, "QlearnIA": 1}, engine=OracleEngine(3))
    q = bg.ships[2].agent
    q.is_learning = False                    # the CPU stand-in engine has no fit: forward only
    assert isinstance(q, qlearn.QlearnIA) and q.trainer is qlearn.TRAINER
    forwards = 0
    for t in range(24):
        bg.frame()
        a = bg.actions[2]
        if a is not None:
            assert int(a.shoot) + int(a.thrust) == 1
            assert 0 <= a.pointing.x <= 399 and 0 <= a.pointing.y <= 399
        if qlearn.TRAINER.act_values is not None:
            forwards += 1
    assert q.total_steps == 24 and forwards >= 1
    bg.restart()
    assert q.done is False and q.previous_obs is None and len(q.scores) == 1


@pytest.mark.gpu
def test_policy_explore_gpu_vs_oracle():
    """Device epsilon-greedy over the forward results == the oracle's restatement of the law, bit for bit."""
    from ofighters_amd import ArenaBatch, DeviceBuffer
    from ofighters_amd.agents.policy_weights import synthetic
    from oracle import pyoracle
    N, M, seed = 6, 5, 99
    b = ArenaBatch(N, M, arena_base=40)
    b.spawn_random(seed)
    w = synthetic()
    base = b.policy_forward_host(w)
    S = N * M
    for eps, collecting, tick in ((0.3, False, 7), (0.0, False, 8), (1.0, False, 9), (0.05, True, 10)):
        di = DeviceBuffer(4 * S).upload(base["iaction"])
        dp = DeviceBuffer(8 * S).upload(base["ipointer"])
        b.policy_explore(eps, seed, tick=tick, collecting=collecting, iaction_ptr=di.ptr, ipointer_ptr=dp.ptr)
        b.sync()
        ia, ip = di.download(np.int32, (N, M)), dp.download(np.int32, (N, M, 2))
        cfg = pyoracle.default_cfg(M)
        hits = 0
        for g in range(N):
            for i in range(M):
                r = pyoracle.policy_explore(cfg, eps, seed, 40 + g, i, tick, collecting)
                if r is None:
                    assert ia[g, i] == base["iaction"][g, i] and tuple(ip[g, i]) == tuple(base["ipointer"][g, i])
                else:
                    hits += 1
                    assert (ia[g, i], ip[g, i, 0], ip[g, i, 1]) == r
                    assert ia[g, i] in (0, 1) and 0 <= ip[g, i, 0] <= 399 and 0 <= ip[g, i, 1] <= 399
        assert hits == (S if (collecting or eps == 1.0) else hits)
        if eps == 0.0 and not collecting:
            assert hits == 0
    b.close()


@pytest.mark.gpu
def test_qlearnia_flow_on_gpu():
    from ofighters_amd.lib.battleground import Battleground
    from ofighters_amd.agents import qlearn
    from ofighters_amd.lib.epsilon import Epsilon_decay
    random.seed(1)
    np.random.seed(1)
    qlearn.TRAINER = qlearn.Trainer(epsilon=Epsilon_decay())
    qlearn.TRAINER.epsilon.set(0.0)           # always the network after the collecting phase
    bg = Battleground(ships={"idle": 6, "QlearnIA": 1})
    bg.ships[6].agent.is_learning = False
    for t in range(23):
        bg.frame()
    a = bg.actions[6]
    assert a is not None and int(a.shoot) + int(a.thrust) == 1
    assert qlearn.TRAINER.act_values is not None and qlearn.TRAINER.act_values.shape == (2,)
