"""CPU stand-in for ArenaBatch built on the oracle (TEST ONLY): lets the host
logic of the sharded rollout run under gloo without a GPU."""
import numpy as np

from oracle import pyoracle
from ofighters_amd import _native as nat


class OracleBatch:
    def __init__(self, n_arenas, n_ships=8, arena_base=0):
        self.N, self.M, self.base = n_arenas, n_ships, arena_base
        self.arenas = [pyoracle.Arena(n_ships=n_ships) for _ in range(n_arenas)]
        self.episode = 0
        self._acts = None
        self._sums = np.zeros(n_ships + 1, np.int64)

    def spawn_random(self, seed):
        for g, a in enumerate(self.arenas):
            a.spawn(pyoracle.reset_draws(a.cfg, seed, self.base + g, 0))

    def restart_random(self, seed):
        self.episode += 1
        self._sums[:] = 0
        for g, a in enumerate(self.arenas):
            self._sums[:self.M] += a.ships()["score"]
            self._sums[self.M] += 1
            a.restart(pyoracle.reset_draws(a.cfg, seed, self.base + g, self.episode))

    def bot_actions(self, behaviours, seed, tick=None):
        beh = np.array([nat.BEHAVIOURS[b] for b in behaviours], np.int32)
        self._acts = [a.bot_actions(beh, seed, self.base + g, tick) for g, a in enumerate(self.arenas)]

    def step(self):
        for a, act in zip(self.arenas, self._acts):
            a.step(act)

    def rasterise(self):
        pass

    def episode_scores(self):
        return self._sums.copy()


class OracleEngine:
    """Single-arena engine with the interface the facade (lib/battleground.py) needs, on the CPU oracle."""

    def __init__(self, n_ships, width=400, height=400):
        self.N, self.M, self.W, self.H = 1, n_ships, width, height
        self.a = pyoracle.Arena(cfg=pyoracle.default_cfg(n_ships, width=width, height=height))

    def spawn(self, draws):
        self.a.spawn(np.asarray(draws, np.int32).reshape(self.M, 2))

    def restart(self, draws):
        self.a.restart(np.asarray(draws, np.int32).reshape(self.M, 2))

    def step_packed(self, packed):
        self.a.step(np.asarray(packed, np.int32).reshape(self.M, 5))

    def snapshot(self):
        s, l = self.a.ships(), self.a.lasers()
        n = len(l["x"])
        one = lambda v: np.asarray(v)[None]
        return dict(x=one(s["xy"][:, 0]), y=one(s["xy"][:, 1]), px=one(s["pt"][:, 0]), py=one(s["pt"][:, 1]),
                    alive=one(s["alive"]), hull=one(s["hull"]), reward=one(s["reward"]), score=one(s["score"]),
                    n_lasers=np.array([n]), lx=one(l["x"]), ly=one(l["y"]), lowner=one(l["owner"]), ldead=one(l["destroyed"]))

    def maps_f64(self):
        sm, lm = self.a.rasterise()
        return sm[None].astype(np.float64), lm[None].astype(np.float64)

    def policy_forward_host(self, weights, ship_mask=None, want_heat=False):
        sm, lm = self.a.rasterise()
        head, _ = self.a.obs_head()
        M = self.M
        out = dict(act=np.zeros((1, M, 2), np.float32), iaction=np.zeros((1, M), np.int32),
                   ipointer=np.zeros((1, M, 2), np.int32))
        for i in range(M):
            if ship_mask is not None and not np.asarray(ship_mask).reshape(M)[i]:
                continue
            act, _, ia, ip = pyoracle.policy_forward(sm, lm, head[i].astype(np.float32), weights, want_heat=False)
            out["act"][0, i], out["iaction"][0, i], out["ipointer"][0, i] = act, ia, ip
        return out
