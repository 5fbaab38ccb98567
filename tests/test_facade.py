"""The facade keeps the reference's Python protocol: with the same random.seed
it consumes the Mersenne-Twister stream in the reference's order, so a whole
reference run (spawn draws, every bot action, every restart draw) is reproduced
from the seed alone.  CPU tests inject the oracle as the engine; the GPU test
runs the same check on libofx."""
import random

import numpy as np
import pytest

from ofighters_amd.lib.action import Action
from ofighters_amd.lib.battleground import Battleground
from ofighters_amd.lib.couple import Couple, Point
from ofighters_amd.lib.observation import Observation
from ofighters_amd.agents.agent import Agent
from tests.trace_util import load_trace, unpack_map

SEEDED = {"random4_s1": (1, {"random": 4}), "random8_s42": (42, {"random": 8}),
          "mixed8_s5": (5, {"random": 3, "turret": 2, "runner": 1, "shoot": 1, "thrust": 1}),
          "turret8_s9": (9, {"turret": 8})}


def replay_seeded(name, engine_factory, setup=None):
    seed, ships = SEEDED[name]
    z = load_trace(name)
    M = z["init_state"].shape[0]
    random.seed(seed)
    bg = Battleground(ships=ships, engine=engine_factory(M))
    if setup is not None:
        setup(bg)
    assert [(s.body.x, s.body.y) for s in bg.ships] == [tuple(r) for r in z["spawn_draws"]]
    ticks, episodes = int(z["ticks"]), int(z["episodes"])
    map_idx = {int(t): k for k, t in enumerate(z["map_ticks"])}
    t = 0
    for ep in range(episodes):
        for _ in range(ticks):
            bg.frame()
            for i, a in enumerate(bg.actions):
                want = z["actions"][t, i]
                got = (0, 0, 0, 0, 0) if a is None else a.packed()
                assert tuple(want) == tuple(got), (name, t, i)
            assert [(s.body.x, s.body.y) for s in bg.ships] == [tuple(r) for r in z["ship_xy"][t]]
            assert [s.is_playable() for s in bg.ships] == [bool(v) for v in z["ship_alive"][t]]
            assert [s.agent.reward for s in bg.ships] == list(z["reward"][t])
            assert [s.agent.score for s in bg.ships] == list(z["score"][t])
            assert len(bg.lasers) == int(z["n_lasers"][t])
            assert [l.body.x for l in bg.lasers] == list(z["laser_x"][t, :len(bg.lasers)])
            if t in map_idx:
                obs = bg.absolute_state
                assert np.array_equal(obs.ship_map, unpack_map(z["ship_maps"][map_idx[t]]))
                assert np.array_equal(obs.laser_map, unpack_map(z["laser_maps"][map_idx[t]]))
                assert obs.ship_map.dtype == np.float64 and obs.ship_map.shape == (400, 400)
            t += 1
        bg.restart()
        assert [[s.body.x, s.body.y, s.pointing.x, s.pointing.y] for s in bg.ships] == z["reset_state"][ep].tolist()
        assert [s.agent.scores[-1] for s in bg.ships] == list(z["ep_scores"][ep])
    return bg


@pytest.mark.parametrize("name", sorted(SEEDED))
def test_facade_reproduces_reference_from_seed_cpu(name):
    from tests.oracle_batch import OracleEngine
    replay_seeded(name, lambda M: OracleEngine(M))


def _qlearnia_collecting_run(engine_factory):
    """The reference's own QlearnIA ships (collecting phase: random_play) next to random bots, from the seed alone;
    afterwards the shared Trainer.memory holds exactly the reference's last 400 remembered rows."""
    from ofighters_amd.agents import qlearn
    qlearn.TRAINER = None
    qlearn.QlearnIA.max_id = 1
    SEEDED["replay_open"] = (11, {"QlearnIA": 3, "random": 2})

    def setup(bg):
        for s in bg.ships:
            if s.agent.behavior == "QlearnIA":
                s.agent.collecting_steps = 10 ** 9       # same harness settings as oracle/gen_golden.py
                s.agent.is_learning = False
    try:
        replay_seeded("replay_open", engine_factory, setup)
        z = load_trace("replay_open")
        mem = list(qlearn.TRAINER.memory)
        ref = z["replay_rows"][-len(mem):]
        assert len(mem) == int(z["replay_len"]) == qlearn.TRAINER.memory.maxlen
        got = [[m[1], m[2][0], m[2][1], m[3], int(bool(m[5]))] for m in mem]
        assert got == ref[:, 2:].tolist()
    finally:
        del SEEDED["replay_open"]
        qlearn.TRAINER = None


def test_facade_qlearnia_memory_reproduces_reference_cpu():
    from tests.oracle_batch import OracleEngine
    _qlearnia_collecting_run(lambda M: OracleEngine(M))


@pytest.mark.gpu
def test_facade_qlearnia_memory_reproduces_reference_gpu():
    from ofighters_amd import ArenaBatch
    _qlearnia_collecting_run(lambda M: ArenaBatch(1, M))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["random8_s42", "mixed8_s5"])
def test_facade_reproduces_reference_from_seed_gpu(name):
    from ofighters_amd import ArenaBatch
    replay_seeded(name, lambda M: ArenaBatch(1, M))


def test_bot_plugin_protocol_and_observation_contract():
    """Any object with .play(obs) plugs in through Agent(bot=...) (agents/agent.py:19-37) and
    sees the reference's Observation attributes (lib/observation.py:50-68)."""
    from tests.oracle_batch import OracleEngine
    seen = []

    class Sniper:
        def play(self, obs):
            seen.append(obs)
            assert isinstance(obs.pointing, Point) and isinstance(obs.pos, Point) and isinstance(obs.dim, Couple)
            assert obs.ship_map.shape == (400, 400) and obs.laser_map.shape == (400, 400)
            assert obs.vector.shape == (320008, 1) and obs.vector.dtype == np.float64
            assert list(obs.vector[:8, 0]) == [obs.reward, 1, obs.pointing.x, obs.pointing.y, 400, 400, obs.pos.x, obs.pos.y]
            assert obs.vector[8:160008, 0].sum() == obs.ship_map.sum()
            if obs.done:
                return None
            return Action(shoot=True, thrust=False, pointing=Point(200, 200))

    random.seed(3)
    bg = Battleground(ships={Sniper(): 1, "idle": 2}, engine=OracleEngine(3))
    for _ in range(5):
        bg.frame()
    assert len(seen) == 5 and len(bg.lasers) >= 1
    assert bg.ships[0].agent.bot is not None and bg.ships[1].agent.behavior == "idle"


def test_error_messages_match_reference():
    with pytest.raises(Exception, match="ships argument must be int or dict."):
        Battleground(ships="eight")                                           # battleground.py:30
    with pytest.raises(Exception, match="pointing argument must be specified."):
        Action(shoot=True)                                                    # action.py:40
    with pytest.raises(Exception, match="Invalid vector"):
        Action(vector=np.zeros((4,)))                                         # action.py:63
    with pytest.raises(Exception, match="select an existing behavior"):
        Agent("kamikaze")                                                     # agent.py:51
    with pytest.raises(Exception, match="analyse_battleground first"):
        Observation().toVector()                                              # observation.py:114-115
    a = Action(vector=np.array([[1.0], [0.0], [12.0], [34.0]]))
    assert a.shoot and not a.thrust and (a.pointing.x, a.pointing.y) == (12.0, 34.0)
    assert Action(shoot=True, pointing=Point(3, 4)).vector.tolist() == [1, 0, 3, 4]
