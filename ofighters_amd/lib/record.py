"""OfighterRecord - record / replay of a game and a headless frame dump.

API mirror of the reference's lib/record.py:7-60 (saveFrame / nextFrame / rewind / save / load) with two deliberate
differences, both because the reference's own path cannot work: its records are pickles of live objects whose replay
calls `game.frame(actions)` - a signature Battleground.frame does not have (lib/battleground.py:163) - and its
`load` is shadowed by a second definition that raises (record.py:62-64).  Here a record is DATA: the spawn draws, the
packed actions of every frame (lib/action.py:12-56 as 5 integers) and the reset draws of every restart, stored as a
numpy `.orec.npz`; the engine is deterministic, so replaying the log through scripted agents reproduces the game
bit for bit.  `render` replaces the Tk canvas (lib/ofighters.py:578-641) with an RGB array / PNG file."""
import struct
import zlib

import numpy as np

from .action import Action
from .battleground import Battleground
from .couple import Point


class _LogBot:
    """bot plug-in (agents/agent.py:34-37 protocol) that plays a recorded column of actions"""

    def __init__(self, record, index):
        self.record, self.index = record, index

    def play(self, obs):
        a = self.record.actions[self.record.reading_head][self.index]
        if not a[0]:
            return None
        return Action(shoot=bool(a[1]), thrust=bool(a[2]), pointing=Point(int(a[3]), int(a[4])))


class OfighterRecord:
    def __init__(self, battleground=None, engine_factory=None):
        self.actions = []              # per frame [M][5]: valid, shoot, thrust, px, py
        self.restarts = []             # (frame index before which Battleground.restart ran, reset draws [M][2])
        self.engine_factory = engine_factory
        self.game = None
        self.reading_head = 0
        if battleground is not None:
            self.spawn_draws = battleground.spawn_draws.copy()
            self.dim = (battleground.dim.x, battleground.dim.y)

    # ---- recording (reference: saveFrame(actions), record.py:24-25)
    def saveFrame(self, actions):
        self.actions.append(np.array([(0, 0, 0, 0, 0) if a is None else a.packed() for a in actions], np.int32))

    def saveRestart(self, battleground):
        self.restarts.append((len(self.actions), battleground.last_reset_draws.copy()))

    # ---- replay (reference: rewind / nextFrame, record.py:28-41)
    def rewind(self):
        M = len(self.spawn_draws)
        engine = self.engine_factory(M) if self.engine_factory else None
        self.game = Battleground(ships={"idle": M}, largeur=self.dim[0], hauteur=self.dim[1], engine=engine,
                                 spawn_draws=self.spawn_draws)
        from ..agents.agent import Agent
        for i, ship in enumerate(self.game.ships):
            ship.agent = Agent(bot=_LogBot(self, i))
        self.reading_head = 0
        self._restart_at = {int(t): d for t, d in self.restarts}

    def nextFrame(self):
        """Advance the replay by one frame and return the new observation (Battleground.absolute_state)."""
        if self.reading_head in self._restart_at:
            self.game.restart(reset_draws=self._restart_at[self.reading_head])
        self.game.frame()
        self.reading_head += 1
        return self.game.absolute_state

    def __str__(self):
        return str(self.actions)

    # ---- files
    def save(self, name):
        if not name.endswith(".orec.npz"):
            name += ".orec.npz"
        M = len(self.spawn_draws)
        np.savez_compressed(name, spawn_draws=self.spawn_draws, dim=np.array(self.dim, np.int32),
                            actions=np.array(self.actions, np.int32).reshape(len(self.actions), M, 5),
                            restart_frames=np.array([t for t, _ in self.restarts], np.int32),
                            restart_draws=np.array([d for _, d in self.restarts], np.int32).reshape(len(self.restarts), M, 2))
        return name

    @classmethod
    def load(cls, name, engine_factory=None):
        if not name.endswith(".orec.npz"):
            name += ".orec.npz"
        z = np.load(name)              # plain arrays: nothing is unpickled
        rec = cls(engine_factory=engine_factory)
        rec.spawn_draws, rec.dim = z["spawn_draws"], tuple(int(v) for v in z["dim"])
        rec.actions = list(z["actions"])
        rec.restarts = list(zip((int(t) for t in z["restart_frames"]), z["restart_draws"]))
        rec.rewind()
        return rec


def render(battleground):
    """uint8 RGB frame [H][W][3]: playable ships white, lasers red on black (the maps of the current observation)."""
    obs = battleground.absolute_state
    ship, laser = obs.ship_map != 0, obs.laser_map != 0
    img = np.zeros(ship.shape + (3,), np.uint8)
    img[ship] = (255, 255, 255)
    img[laser & ~ship] = (255, 40, 40)
    return img


def save_png(path, rgb):
    """Minimal PNG writer (8-bit RGB, no dependency beyond zlib)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    return path
