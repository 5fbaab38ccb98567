"""Observation - what a bot sees (API mirror of the reference's
lib/observation.py:14-151).

Battleground-level part (`analyse_battleground`): `dim`, `ship_map`,
`laser_map` - float64 (dim.x, dim.y) arrays indexed [y][x], rasterised on the
device (ofx_rasterise).  Ship-level part (`analyse_ship`): `reward`,
`can_shoot`, `pointing`, `pos`, `done`, and `vector` = the (320008, 1) float64
column [reward, can_shoot, pointing.x, pointing.y, dim.x, dim.y, pos.x, pos.y,
ship_map.ravel(), laser_map.ravel()] (observation.py:119-125) - materialised
lazily here because copying 2.56 MB per ship per tick is 61-68 % of the
reference's CPU time and most bots never read the tail.
"""
import numpy as np

from .couple import Couple, Point

DEFAULT_WIDTH = 400
DEFAULT_HEIGHT = 400


class Observation:
    observations = {"can_shoot": 1, "reward": 1, "pointing": 2, "dim": 2, "pos": 2,
                    "ships_map": DEFAULT_WIDTH * DEFAULT_HEIGHT, "lasers_map": DEFAULT_WIDTH * DEFAULT_HEIGHT}
    size = sum(observations.values())

    def __init__(self, **kwargs):
        self._vector = None
        self.btlgA = False
        self.shipA = False
        self.battleground = None
        self.dim = None
        self.ship_map = None
        self.laser_map = None
        self.reward = None
        self.can_shoot = None
        self.pointing = None
        self.pos = None
        self.done = None
        self._ship = None
        battleground = kwargs.get("battleground")
        ship = kwargs.get("ship")
        if battleground:
            self.analyse_battleground(battleground)
        if ship:
            self.analyse_ship(ship)

    def analyse_battleground(self, battleground):
        self.battleground = battleground
        self.dim = battleground.dim
        self.ship_map, self.laser_map = battleground._maps()
        self.btlgA = True

    def analyse_ship(self, ship):
        """Must be executed after analyse_battleground"""
        self._ship = ship
        self.reward = ship.agent.reward
        self.can_shoot = 0 if ship.can_shoot == 0 else 1
        self.pointing = ship.pointing
        self.pos = Point(ship.body.x, ship.body.y)
        self.shipA = True
        self.done = not ship.is_playable()
        self._vector = None
        self._check()

    def for_ship(self, ship):
        """A per-ship view sharing the maps (the reference mutates ONE object per ship,
        battleground.py:150; a fresh view keeps retained observations stable)."""
        view = Observation()
        view.battleground, view.dim = self.battleground, self.dim
        view.ship_map, view.laser_map, view.btlgA = self.ship_map, self.laser_map, self.btlgA
        view.analyse_ship(ship)
        return view

    def _check(self):
        if not self.btlgA:
            raise Exception("You must execute analyse_battleground first.")
        if not self.shipA:
            raise Exception("You must execute analyse_ship first.")

    def head(self):
        return np.array([self.reward, self.can_shoot, self.pointing.x, self.pointing.y, self.dim.x, self.dim.y,
                         self.pos.x, self.pos.y], dtype=np.float64)

    def toVector(self):
        self._check()
        parts = (self.head(), np.asarray(self.ship_map, np.float64).ravel(), np.asarray(self.laser_map, np.float64).ravel())
        self._vector = np.concatenate(parts).reshape(-1, 1)
        return self._vector

    @property
    def vector(self):
        if self._vector is None and self.btlgA and self.shipA:
            self.toVector()
        return self._vector

    def _vector_or_none(self):
        """Agent.step stores obs.vector (agent.py:76); do not force the 2.56 MB copy for that."""
        return self._vector if self._vector is not None else np.array([])

    def fromVector(self, vector):
        raise Exception("Not implemented.")

    def toBattleground(self):
        raise Exception("Not implemented.")
