"""Battleground - the world container and its tick, as a facade over one arena
of a device batch (API mirror of the reference's lib/battleground.py:10-173 and
of the Ship / Laser attributes its callers read, lib/ship.py:32-110,
lib/laser.py:16-34).

    bg = Battleground(ships={"random": 8})      # same constructor arguments
    bg.frame()                                   # request_actions -> generate_frame -> Observation
    bg.restart()

The per-tick arithmetic (laser advance, hit tests, thrust, spawn, rewards,
rasterisation) runs in libofx on the GPU; this module keeps the reference's
Python protocol around it: every ship - dead ones too - has `agent.step(obs)`
called in index order (battleground.py:146-150, ship.py:253-282), the returned
Action objects (or None) are packed into ofx_action records, and the laser list
is cleaned of last tick's destroyed lasers the way the GUI does between ticks
(lib/ofighters.py:619-625,702-707 - inside ofx_step).
"""
from random import randint

import numpy as np

from .couple import Couple, Point
from .observation import DEFAULT_HEIGHT, DEFAULT_WIDTH, Observation
from ..agents.agent import Agent

SHIPS_SPEED = 8          # lib/ship.py:24
REWARDS = {"death": 0, "kill": 0, "aim": 2, "trajectory": 1}   # agents/qlearnIA_V2.py:39-44


class _Body:
    """Circle(x, y, radius) view (lib/form.py:142-146)."""

    def __init__(self, x, y, radius):
        self.x, self.y, self.radius = x, y, radius


class Ship:
    """Read-mostly view of one ship slot; mirrors the attributes callers use (lib/ship.py:35-110)."""
    id_max = 1

    def __init__(self, x, y, battleground, behavior="idle", bot=None):
        self.id = Ship.id_max
        Ship.id_max += 1
        self.time = 0
        self.body = _Body(x, y, 8)
        self.hull = 1
        self.battleground = battleground
        self.max_speed = SHIPS_SPEED
        self.pointing = Point(x, y)
        self.state = "flying"
        self.can_shoot = 1
        self.player = None
        self.laser_speed = 10
        if behavior == "QlearnIA":          # lib/ship.py:69-74
            from ..agents.qlearn import QlearnIA
            self.agent = QlearnIA()
        else:
            self.agent = Agent(behavior, bot=bot)

    def is_playable(self):
        return self.state not in ["destroyed", "wreckage"]

    def is_me(self, ship):
        return self.id == ship.id

    def get_action(self, obs):
        """ship.py:253-282: the agent is stepped even when the ship is dead; its action is then dropped."""
        view = obs.for_ship(self)
        if view.done:
            self.agent.step(view)
            return None
        action = self.agent.step(view)
        return action if action else None


class Laser:
    """Read-only view of one laser of the list (lib/laser.py:16-34)."""

    def __init__(self, x, y, owner, destroyed):
        self.body = _Body(x, y, 2)
        self.owner = owner
        self.state = "destroyed" if destroyed else "flying"


class Battleground:
    def __init__(self, state=None, ships=2, largeur=DEFAULT_WIDTH, hauteur=DEFAULT_HEIGHT, networks=[],
                 engine=None, laser_cap=512, spawn_draws=None):
        default_behavior = "random"
        if isinstance(ships, dict):
            self.ships_map = ships
        elif isinstance(ships, int):
            self.ships_map = {default_behavior: ships}
        else:
            raise Exception("ships argument must be int or dict.")
        if state:
            raise Exception("Not implemented.")   # Observation.loadBattleground does not exist in the reference either
        self.ships_number = len(self.ships_map)
        self.time = 0
        self.dim = Couple(largeur, hauteur)
        self.ships = []
        self.lasers = []
        self.networks = networks
        self.actions = []
        draws = []
        for behavior, number in self.ships_map.items():
            for _ in range(number):
                if spawn_draws is not None:                        # replaying a record: the logged draws
                    x, y = (int(v) for v in spawn_draws[len(draws)])
                else:
                    x, y = randint(0, largeur), randint(0, hauteur)   # battleground.py:81 (inclusive upper bound)
                draws.append((x, y))
                bot = behavior if hasattr(behavior, "play") else None
                self.ships.append(Ship(x, y, self, behavior=None if bot else behavior, bot=bot))
        M = len(self.ships)
        if engine is None:
            from ..engine import ArenaBatch
            engine = ArenaBatch(1, M, width=largeur, height=hauteur, laser_cap=laser_cap)
        self._e = engine
        self.spawn_draws = np.array(draws, np.int32).reshape(M, 2)
        self._e.spawn(self.spawn_draws.reshape(1, M, 2))
        self._maps_cache = None
        self._pull()
        self.absolute_state = Observation(battleground=self)

    # ------------------------------------------------------------------ engine <-> views
    def _maps(self):
        if self._maps_cache is None:
            sm, lm = self._e.maps_f64()
            self._maps_cache = (sm[0], lm[0])
        return self._maps_cache

    def _pull(self):
        """Refresh the Python views from device state after a step / restart."""
        st = self._e.snapshot()
        for i, ship in enumerate(self.ships):
            ship.body.x, ship.body.y = int(st["x"][0, i]), int(st["y"][0, i])
            px, py = int(st["px"][0, i]), int(st["py"][0, i])
            if (ship.pointing.x, ship.pointing.y) != (px, py):
                ship.pointing = Point(px, py)
            alive = bool(st["alive"][0, i])
            if alive:
                ship.state = "flying"
            elif ship.state == "flying":
                ship.state = "destroyed"
            ship.hull = int(st["hull"][0, i])
            ship.agent.reward = int(st["reward"][0, i])
            ship.agent.score = int(st["score"][0, i])
        n = int(st["n_lasers"][0])
        self.lasers = [Laser(float(st["lx"][0, j]), float(st["ly"][0, j]), self.ships[int(st["lowner"][0, j])],
                             bool(st["ldead"][0, j])) for j in range(n)]
        self._maps_cache = None

    def _policy_forward(self, ship, weights):
        """One bi-head forward for `ship` (batch of one, like model.predict in qlearnIA_V2.py:210)."""
        M = len(self.ships)
        mask = np.zeros((1, M), np.uint8)
        i = self.ships.index(ship)
        mask[0, i] = 1
        out = self._e.policy_forward_host(weights, ship_mask=mask)
        return dict(act=out["act"][0, i], iaction=out["iaction"][0, i], ipointer=out["ipointer"][0, i])

    # ------------------------------------------------------------------ reference API
    def outside(self, x, y):
        return (x < 0) or (y < 0) or (x >= self.dim.x) or (y >= self.dim.y)

    def request_actions(self):
        return [ship.get_action(self.absolute_state) for ship in self.ships]

    def generate_frame(self, actions):
        self.time += 1
        M = len(self.ships)
        packed = np.zeros((1, M, 5), np.int32)
        for i, a in enumerate(actions):
            if a is not None:
                packed[0, i] = a.packed()
        self._e.step_packed(packed)
        for ship in self.ships:
            ship.time += 1
        self._pull()

    def frame(self):
        self.actions = self.request_actions()
        self.generate_frame(self.actions)
        self.absolute_state = Observation(battleground=self)

    def restart(self, reset_draws=None):
        self.time = 0
        self.actions = []
        M = len(self.ships)
        draws = np.zeros((1, M, 2), np.int32)
        for i, ship in enumerate(self.ships):
            ship.agent.reset()
            ship.time = 0
            if reset_draws is not None:                                      # replaying a record
                draws[0, i] = reset_draws[i]
            else:
                draws[0, i] = (randint(0, self.dim.x), randint(0, self.dim.y))   # battleground.py:115
        self.last_reset_draws = draws[0].copy()
        self._e.restart(draws)
        self._pull()
        for ship in self.ships:
            ship.state = "flying"
        self.absolute_state = Observation(battleground=self)

    def run(self, ticks=None):
        t = 0
        while ticks is None or t < ticks:
            self.frame()
            t += 1
