#!/opt/conda/bin/python3.9
"""Golden-vector generator (TEST INFRASTRUCTURE - runs only in the build container).

Imports the *live* reference from /root/reference (read-only) under
/opt/conda/bin/python3.9 (numpy 1.26.4, scikit-image 0.18.3) and records small
input/output fixtures for the hot path into tests/golden/*.npz:

  step_<name>.npz   per-tick traces of Battleground.frame()  (battleground.py:163-166)
                    with the GUI laser clean-up of ofighters.py:619-625,702-707
                    emulated between ticks and Battleground.restart() between
                    episodes (ofighters.py:684-688, battleground.py:108-117)
  raster_cases.npz  Circle.binary_draw -> skimage.draw.disk   (form.py:222-228)
  geometry.npz      Ship.thrust / Circle.edge / enemy_aimed / enemy_on_trajectory
                    on random integer configurations          (ship.py:134-222)
  scratch_nn.npz    Neural_network.feed                       (neural_network.py:396-420)
  epsilon.npz       Epsilon_cos / Epsilon_decay sequences     (lib/epsilon.py:36-86)

keras / tensorflow are absent from every interpreter in this image and are not
on the recorded path (scripted bots only); they are replaced by inert
MagicMock modules *for the import only* (ship.py:22 -> qlearnIA_V2.py:6 imports
keras at module load to reach the REWARDS dict).  Nothing from the mocks
reaches a recorded value.  The Keras bi-head forward (P1) is therefore NOT
covered here: its parity is unpinned.

Only data (inputs + expected outputs) is written; no reference source is copied.
Usage:  /opt/conda/bin/python3.9 oracle/gen_golden.py [outdir]
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
from unittest.mock import MagicMock

for _m in ["keras", "keras.models", "keras.layers", "keras.layers.core",
           "keras.optimizers", "keras.layers.advanced_activations",
           "keras.backend", "tensorflow"]:
    sys.modules[_m] = MagicMock()
sys.path.insert(0, "/root/reference")

import random

import numpy as np

_sink = io.StringIO()
with contextlib.redirect_stdout(_sink):
    import ofighters.lib.battleground as bgmod
    from ofighters.lib.battleground import Battleground
    from ofighters.lib.observation import Observation
    from ofighters.lib.action import Action
    from ofighters.lib.couple import Point
    from ofighters.lib.form import Circle
    from ofighters.lib.ship import Ship
    from ofighters.agents.agent import Agent
    from ofighters.agents.neural_network import Neural_network

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

EPISODE_TICKS = 200  # ofighters.py:59 MAX_TIME


# --------------------------------------------------------------------------
# harness-side instrumentation (records values, never changes behaviour)
# --------------------------------------------------------------------------
class RandintTap:
    """Replaces the `randint` name inside ofighters.lib.battleground so that the
    spawn / reset draws (battleground.py:81,115) are recorded as INPUTS.
    lo/hi override lets a scenario confine spawns to a small box."""

    def __init__(self):
        self.log = []
        self.box = None

    def __call__(self, a, b):
        if self.box is not None:
            a, b = self.box
        v = random.randint(a, b)
        self.log.append(v)
        return v


TAP = RandintTap()
bgmod.randint = TAP


class ScriptBot:
    """bot plug-in (agent.py:34-37 protocol: object with .play(obs)).
    script[t] = None | (shoot, thrust, px, py) ; px<0 means keep obs.pointing."""

    def __init__(self, script):
        self.script = script
        self.t = 0

    def play(self, obs):
        a = self.script[self.t] if self.t < len(self.script) else None
        self.t += 1
        if a is None:
            return None
        shoot, thrust, px, py = a
        pointing = obs.pointing if px is None else Point(px, py)
        return Action(shoot=bool(shoot), thrust=bool(thrust), pointing=pointing)


def tap_agent(ship, store):
    """Record obs.vector[:8] and obs.done as seen by Agent.step (agent.py:66)."""
    orig = ship.agent.step

    def step(obs):
        store.append((np.array(obs.vector[:8, 0], dtype=np.float64), bool(obs.done)))
        return orig(obs)

    ship.agent.step = step


def pack(m):
    return np.packbits(np.asarray(m) != 0)


def run_trace(name, ships, seed, episodes, ticks=EPISODE_TICKS, map_every=1,
              init=None, scripts=None, box=None, lmax=None, setup=None, clock=None, extra=None):
    """Drive the reference exactly as the GUI does and dump a trace."""
    random.seed(seed)
    TAP.log = []
    TAP.box = box
    with contextlib.redirect_stdout(_sink):
        bg = Battleground(ships=ships)
    M = len(bg.ships)
    spawn_draws = np.array(TAP.log, dtype=np.int32).reshape(M, 2)
    if init is not None:
        # crafted start: positions / pointing set by hand, then the absolute
        # observation rebuilt with the reference's own class
        for s, (x, y, px, py) in zip(bg.ships, init):
            s.body.x, s.body.y = x, y
            s.pointing = Point(px, py)
        with contextlib.redirect_stdout(_sink):
            bg.absolute_state = Observation(battleground=bg)
    if scripts is not None:
        for s, sc in zip(bg.ships, scripts):
            s.agent = Agent(bot=ScriptBot(sc))
    if setup is not None:
        setup(bg)
    seen = [[] for _ in range(M)]
    for s, st in zip(bg.ships, seen):
        tap_agent(s, st)

    T = episodes * ticks
    init_state = np.array([[s.body.x, s.body.y, s.pointing.x, s.pointing.y]
                           for s in bg.ships], dtype=np.int32)
    init_ship_map = pack(bg.absolute_state.ship_map)
    actions = np.zeros((T, M, 5), dtype=np.int32)       # valid, shoot, thrust, px, py
    obs8 = np.zeros((T, M, 8), dtype=np.float64)
    obs_done = np.zeros((T, M), dtype=np.uint8)
    ship_xy = np.zeros((T, M, 2), dtype=np.int32)
    ship_pt = np.zeros((T, M, 2), dtype=np.int32)
    ship_alive = np.zeros((T, M), dtype=np.uint8)
    reward = np.zeros((T, M), dtype=np.int64)
    score = np.zeros((T, M), dtype=np.int64)
    n_lasers = np.zeros((T,), dtype=np.int32)
    las = []                                            # per tick (n,4): x,y,owner,destroyed
    map_ticks, ship_maps, laser_maps = [], [], []
    reset_draws = np.zeros((episodes, M, 2), dtype=np.int32)
    reset_state = np.zeros((episodes, M, 4), dtype=np.int32)
    ep_scores = np.zeros((episodes, M), dtype=np.int64)
    id2idx = {s.id: i for i, s in enumerate(bg.ships)}

    t = 0
    with contextlib.redirect_stdout(_sink):
        for ep in range(episodes):
            for _ in range(ticks):
                # GUI clean-up happens at the start of the next GUI tick,
                # before Battleground.frame (ofighters.py:663-667)
                bg.lasers = [l for l in bg.lasers if l.state != "destroyed"]
                if clock is not None:
                    clock(t)
                bg.frame()
                for i, s in enumerate(bg.ships):
                    a = bg.actions[i]
                    if a is not None:
                        actions[t, i] = (1, int(a.shoot), int(a.thrust),
                                         int(a.pointing.x), int(a.pointing.y))
                    o, d = seen[i][t]
                    obs8[t, i] = o
                    obs_done[t, i] = d
                    ship_xy[t, i] = (s.body.x, s.body.y)
                    ship_pt[t, i] = (s.pointing.x, s.pointing.y)
                    ship_alive[t, i] = s.is_playable()
                    reward[t, i] = s.agent.reward
                    score[t, i] = s.agent.score
                n_lasers[t] = len(bg.lasers)
                las.append(np.array([[l.body.x, l.body.y, id2idx[l.owner.id],
                                      l.state == "destroyed"] for l in bg.lasers],
                                    dtype=np.float64).reshape(-1, 4))
                if t % map_every == 0 or t % ticks == ticks - 1:
                    map_ticks.append(t)
                    ship_maps.append(pack(bg.absolute_state.ship_map))
                    laser_maps.append(pack(bg.absolute_state.laser_map))
                t += 1
            bg.lasers = [l for l in bg.lasers if l.state != "destroyed"]
            TAP.log = []
            bg.restart()
            reset_draws[ep] = np.array(TAP.log, dtype=np.int32).reshape(M, 2)
            for i, s in enumerate(bg.ships):
                reset_state[ep, i] = (s.body.x, s.body.y, s.pointing.x, s.pointing.y)
                ep_scores[ep, i] = s.agent.scores[-1]
    L = max(1, max(len(a) for a in las))
    if lmax is not None:
        assert L <= lmax, (name, L)
    lx = np.zeros((T, L)); ly = np.zeros((T, L))
    lo = np.zeros((T, L), dtype=np.int16); ld = np.zeros((T, L), dtype=np.uint8)
    for k, a in enumerate(las):
        n = len(a)
        lx[k, :n] = a[:, 0]; ly[k, :n] = a[:, 1]
        lo[k, :n] = a[:, 2]; ld[k, :n] = a[:, 3]
    np.savez_compressed(
        os.path.join(OUT, "step_%s.npz" % name),
        ticks=np.int32(ticks), episodes=np.int32(episodes),
        spawn_draws=spawn_draws, init_state=init_state, init_ship_map=init_ship_map,
        actions=actions, obs8=obs8, obs_done=obs_done,
        ship_xy=ship_xy, ship_pt=ship_pt, ship_alive=ship_alive,
        reward=reward, score=score, n_lasers=n_lasers,
        laser_x=lx, laser_y=ly, laser_owner=lo, laser_destroyed=ld,
        map_ticks=np.array(map_ticks, dtype=np.int32),
        ship_maps=np.array(ship_maps), laser_maps=np.array(laser_maps),
        reset_draws=reset_draws, reset_state=reset_state, ep_scores=ep_scores,
        **(extra() if extra is not None else {}))
    kills = int((ship_alive[ticks - 1::ticks] == 0).sum())
    print("step_%-22s M=%d T=%d maxL=%d deaths=%d reward_sum=%d" % (
        name, M, T, L, kills, int(ep_scores.sum())))


# --------------------------------------------------------------------------
# scenarios
# --------------------------------------------------------------------------
def random_traces():
    # config 1 of BASELINE.json: 1 arena, 4 ships, 200-iter episode, random bot
    for seed in (1, 7, 42, 1234):
        run_trace("random4_s%d" % seed, {"random": 4}, seed, episodes=2, map_every=20)
        run_trace("random8_s%d" % seed, {"random": 8}, seed, episodes=2, map_every=20)
    run_trace("mixed8_s5", {"random": 3, "turret": 2, "runner": 1, "shoot": 1, "thrust": 1},
              5, episodes=2, map_every=25)
    run_trace("turret8_s9", {"turret": 8}, 9, episodes=1, map_every=40)


def brawl_script(rs, M, T, box):
    """Scripted close-quarters fight: pointing often lands on / near the box so
    hits, multi-kills, aim (+2) and trajectory (+1) rewards all occur."""
    lo, hi = box
    scripts = []
    for _ in range(M):
        sc = []
        for _t in range(T):
            r = rs.rand()
            if r < 0.05:
                sc.append(None)
                continue
            shoot = rs.rand() < 0.6
            thrust = rs.rand() < 0.4
            if rs.rand() < 0.7:
                px, py = int(rs.randint(lo - 12, hi + 13)), int(rs.randint(lo - 12, hi + 13))
                px = min(400, max(0, px)); py = min(400, max(0, py))
            else:
                px, py = None, None
            sc.append((shoot, thrust, px, py))
        scripts.append(sc)
    return scripts


def brawl_traces():
    for seed, M, box in ((3, 8, (150, 210)), (11, 8, (0, 50)), (23, 6, (360, 400)), (31, 8, (180, 215))):
        rs = np.random.RandomState(seed)
        eps, ticks = 6, 30
        scripts = brawl_script(rs, M, eps * ticks, box)
        run_trace("brawl%d_s%d" % (M, seed), {"idle": M}, seed, episodes=eps, ticks=ticks,
                  map_every=1, scripts=scripts, box=box)


def crafted_traces():
    K = None  # keep pointing
    idle = (0, 0, K, K)
    # 1. one laser kills two ships in the same tick (no `break`, laser.py:52-60);
    #    ships 1,2 overlap, ship 0 shoots along +x
    init = [(100, 100, 200, 100), (138, 104, 138, 104), (137, 95, 137, 95), (300, 300, 300, 300)]
    scripts = [[(1, 0, K, K)] + [idle] * 9, [idle] * 10, [idle] * 10, [idle] * 10]
    run_trace("craft_multikill", {"idle": 4}, 0, 1, ticks=10, init=init, scripts=scripts)
    # 2. two lasers reach the same target in the same tick: lower list index is
    #    credited, the later one flies through the wreck
    init = [(100, 200, 200, 200), (300, 200, 200, 200), (200, 200, 200, 200), (20, 20, 20, 20)]
    scripts = [[(1, 0, K, K)] + [idle] * 14, [(1, 0, K, K)] + [idle] * 14, [idle] * 15, [idle] * 15]
    run_trace("craft_twolasers", {"idle": 4}, 0, 1, ticks=15, init=init, scripts=scripts)
    # 3. zero-velocity laser (spawn == pointing, laser.py:43) later run into by
    #    its owner (no owner exclusion, laser.py:52-55); also dist==0 shoot (no
    #    laser, form.py:183) and dist==0 thrust (no move, ship.py:218)
    init = [(100, 100, 90, 99), (250, 250, 250, 250), (300, 50, 300, 50)]
    scripts = [[(1, 0, K, K), (0, 0, 40, 97), (0, 1, K, K), (0, 1, K, K), idle, idle],
               [(1, 1, K, K)] * 6, [idle] * 6]
    run_trace("craft_zerovel_selfhit", {"idle": 3}, 0, 1, ticks=6, init=init, scripts=scripts)
    # 4. pointing inside own hit-box: fired = ship centre (ship.py:147-148)
    init = [(200, 200, 203, 204), (215, 220, 215, 220), (50, 350, 50, 350)]
    scripts = [[(1, 0, K, K)] * 3 + [idle] * 5, [idle] * 8, [idle] * 8]
    run_trace("craft_inside_hitbox", {"idle": 3}, 0, 1, ticks=8, init=init, scripts=scripts)
    # 5. thrust clamps (ship.py:221-222), spawn at 400 (battleground.py:81),
    #    lasers leaving through every border (battleground.py:125-126)
    init = [(400, 400, 400, 0), (3, 3, 0, 0), (396, 2, 400, 0), (2, 397, 0, 400), (200, 5, 200, 0)]
    scripts = [[(1, 1, K, K)] * 6, [(1, 1, K, K)] * 6, [(1, 1, K, K)] * 6,
               [(1, 1, K, K)] * 6, [(1, 0, K, K)] * 6]
    run_trace("craft_borders", {"idle": 5}, 0, 1, ticks=6, init=init, scripts=scripts)
    # 6. aim (+2), trajectory (+1), both (+3); cone straddling 0/2pi is never a
    #    hit (ship.py:203-208: no wrap handling) - target due -x of the shooter
    init = [(200, 200, 100, 200), (100, 200, 100, 200),      # target at angle 0 (+pi shift): straddle
            (200, 300, 260, 300), (260, 300, 260, 300),      # target due +x : aim + trajectory
            (50, 50, 50, 120), (52, 90, 52, 90)]             # target nearly on line: trajectory only
    scripts = [[(1, 0, K, K)] * 2 + [idle] * 2, [idle] * 4, [(1, 0, K, K)] * 2 + [idle] * 2, [idle] * 4,
               [(1, 0, K, K)] * 2 + [idle] * 2, [idle] * 4]
    run_trace("craft_rewards", {"idle": 6}, 0, 1, ticks=4, init=init, scripts=scripts)
    # 7. ordered ship loop: ship 0 moves first, ship 1's aim test then sees the
    #    NEW position of ship 0 and the OLD position of ship 2 (ship.py:158-161)
    init = [(100, 100, 140, 100), (300, 300, 108, 100), (200, 100, 160, 100), (300, 200, 200, 100)]
    scripts = [[(0, 1, K, K)] * 3, [(1, 0, K, K)] * 3, [(0, 1, K, K)] * 3, [(1, 0, K, K)] * 3]
    run_trace("craft_order", {"idle": 4}, 0, 1, ticks=3, init=init, scripts=scripts)
    # 8. reset quirks (ship.py:99-101): pointing = OLD position, `x or old`
    #    keeps the coordinate when the draw is 0 -> confine draws to {0,1}
    run_trace("craft_reset_zero", {"random": 4}, 2, 4, ticks=5, box=(0, 1))


# --------------------------------------------------------------------------
# raster / geometry / scratch-NN vectors
# --------------------------------------------------------------------------
def raster_cases():
    cases = []
    rs = np.random.RandomState(77)
    # integer centres, radius 8 (ships) incl. every border/corner and x|y == 400
    for x, y in ((200, 200), (0, 0), (399, 399), (400, 400), (400, 0), (0, 400), (7, 392),
                 (8, 8), (391, 391), (3, 200), (200, 396)):
        cases.append((float(x), float(y), 8.0))
    # lasers: integer and fractional centres, in / straddling / outside the map
    for x, y in ((100.0, 50.0), (100.5, 50.25), (0.0, 0.0), (-1.0, 5.0), (-2.0, 5.0), (-1.999, 5.0),
                 (-2.5, 5.0), (399.0, 399.0), (401.0, 200.0), (401.999, 200.0), (402.0, 200.0),
                 (200.0, 401.5), (200.0, -1.5), (412.3, 77.7), (-9.0, -9.0), (1e-9, 399.999999)):
        cases.append((x, y, 2.0))
    for _ in range(40):
        cases.append((float(rs.uniform(-4, 404)), float(rs.uniform(-4, 404)), 2.0))
    for _ in range(12):
        cases.append((float(rs.randint(0, 401)), float(rs.randint(0, 401)), 8.0))
    # accumulated (non-representable) laser coordinates like the step produces
    for _ in range(24):
        x0, y0 = rs.randint(0, 400, size=2)
        dx, dy = rs.randint(-50, 51, size=2)
        d = np.sqrt(float(dx * dx + dy * dy)) or 1.0
        x, y = float(x0), float(y0)
        for _k in range(rs.randint(1, 30)):
            x += dx * 10 / d
            y += dy * 10 / d
        cases.append((x, y, 2.0))
    cases = np.array(cases, dtype=np.float64)
    maps = []
    for x, y, r in cases:
        # ships carry Python ints, lasers floats: keep that distinction
        cx, cy = (int(x), int(y)) if r == 8.0 else (float(x), float(y))
        g = Circle(cx, cy, int(r)).binary_draw(np.zeros((400, 400)))
        maps.append(pack(g))
    np.savez_compressed(os.path.join(OUT, "raster_cases.npz"), cases=cases, maps=np.array(maps))
    print("raster_cases: %d discs, pixel counts %s ..." % (
        len(cases), [int(np.unpackbits(m).sum()) for m in maps[:12]]))


def geometry_cases(n=20000):
    rs = np.random.RandomState(123)
    with contextlib.redirect_stdout(_sink):
        bg = Battleground(ships={"idle": 2})
    me, foe = bg.ships
    inp = np.zeros((n, 6), dtype=np.int32)     # sx, sy, px, py, ex, ey
    thrust = np.zeros((n, 2), dtype=np.int32)
    edge = np.zeros((n, 3), dtype=np.int32)    # ok, ex, ey
    aimed = np.zeros((n,), dtype=np.uint8)
    traj = np.zeros((n,), dtype=np.uint8)
    fired_centre = np.zeros((n,), dtype=np.uint8)
    for k in range(n):
        mode = k % 4
        sx, sy = rs.randint(0, 401, size=2)
        if mode == 0:      # far field
            px, py = rs.randint(0, 401, size=2); ex, ey = rs.randint(0, 401, size=2)
        elif mode == 1:    # enemy close to the pointing
            ex, ey = rs.randint(0, 401, size=2)
            px, py = ex + rs.randint(-12, 13), ey + rs.randint(-12, 13)
        elif mode == 2:    # everything close (cone wide, hit-box cases, dist 0)
            px, py = sx + rs.randint(-14, 15), sy + rs.randint(-14, 15)
            ex, ey = sx + rs.randint(-30, 31), sy + rs.randint(-30, 31)
        else:              # axis-aligned / straddling 0|2pi
            px, py = sx - rs.randint(0, 200), sy + rs.randint(-3, 4)
            ex, ey = sx - rs.randint(1, 200), sy + rs.randint(-6, 7)
        px, py, ex, ey = [int(min(400, max(0, v))) for v in (px, py, ex, ey)]
        sx, sy = int(sx), int(sy)
        inp[k] = (sx, sy, px, py, ex, ey)
        me.body.x, me.body.y = sx, sy
        me.pointing = Point(px, py)
        foe.body.x, foe.body.y = ex, ey
        aimed[k] = me.enemy_aimed(me.pointing, foe)
        traj[k] = me.enemy_on_trajectory(me.pointing, foe)
        e = me.body.edge(px, py, 2)
        edge[k] = (0, 0, 0) if e is None else (1, e[0], e[1])
        fired_centre[k] = me.body.collide(Circle(px, py, 2))
        me.thrust()
        thrust[k] = (me.body.x, me.body.y)
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), inp=inp, thrust=thrust, edge=edge,
                        aimed=aimed, traj=traj, fired_centre=fired_centre)
    print("geometry: n=%d aimed=%d traj=%d edge_none=%d" % (n, aimed.sum(), traj.sum(), (edge[:, 0] == 0).sum()))


def scratch_nn_cases():
    """Weights are regenerated by the test from the RandomState seed (legacy
    RandomState streams are frozen across numpy versions); only inputs and the
    reference outputs are stored."""
    out = {}
    for tag, layers, seed, nin in (("a", [8, 9, 4], 0, 16), ("b", [64, 32, 16, 4], 1, 8), ("c", [5000, 9, 4], 2, 4)):
        np.random.seed(seed)
        nn = Neural_network(list(layers))
        rs = np.random.RandomState(1000 + seed)
        X = rs.uniform(-2, 2, size=(nin, layers[0]))
        if tag == "c":
            X = (rs.rand(nin, layers[0]) < 0.02).astype(np.float64)   # sparse binary like the map tail
        Y = np.stack([nn.feed(x.reshape(-1, 1))[:, 0] for x in X])
        out["layers_" + tag] = np.array(layers, dtype=np.int32)
        out["seed_" + tag] = np.int32(seed)
        out["x_" + tag] = X
        out["y_" + tag] = Y
        out["argmax_" + tag] = np.array([Neural_network.max_sol_index(y) for y in Y], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "scratch_nn.npz"), **out)
    print("scratch_nn:", {k: v.shape for k, v in out.items() if k.startswith("y_")})


def replay_traces():
    """Trainer.remember / QlearnIA.play bookkeeping (qlearnIA_V2.py:58,237-238,360-403): QlearnIA ships kept in
    their collecting phase (random_play, :393-395 - the Keras model is a mock and is never called) next to random
    bots; every remember() call is recorded in order together with the lock-step and ship it came from."""
    import ofighters.agents.qlearnIA_V2 as ql

    for name, ships, seed, box in (("replay_open", {"QlearnIA": 3, "random": 2}, 11, None),
                                   ("replay_brawl", {"random": 1, "QlearnIA": 4}, 12, (150, 250))):
        ctx = {"t": 0, "ship": -1}
        rows = []
        ql.TRAINER.memory.clear()

        def remember(state, iaction, ipointer, reward, next_state, done, _orig=ql.TRAINER.remember.__func__):
            rows.append((ctx["t"], ctx["ship"], int(iaction), int(ipointer[0]), int(ipointer[1]), int(reward),
                         int(bool(done)), np.array(next_state.vector[:8, 0], dtype=np.float64)))
            _orig(ql.TRAINER, state, iaction, ipointer, reward, next_state, done)

        ql.TRAINER.remember = remember
        chosen = []  # (tick, ship, iaction, px, py) for every play() that chose an action

        def setup(bg):
            for i, s in enumerate(bg.ships):
                if s.agent.behavior != "QlearnIA":
                    continue
                s.agent.collecting_steps = 10 ** 9   # harness: stay in the collecting phase
                s.agent.is_learning = False          # harness: no replay()/fit on the mock model

                def play(obs, _i=i, _a=s.agent, _orig=s.agent.play):
                    ctx["ship"] = _i
                    was_done = _a.done
                    out = _orig(obs)
                    if not was_done:
                        chosen.append((ctx["t"], _i, int(_a.previous_action), int(_a.previous_pointer[0]),
                                       int(_a.previous_pointer[1])))
                    return out

                s.agent.bot_play = play

        def clock(t):
            ctx["t"] = t

        def extra():
            mem = list(ql.TRAINER.memory)
            tail = rows[-len(mem):]
            for m, r in zip(mem, tail):   # the deque holds exactly the last maxlen remembered rows, in order
                assert (m[1], m[2][0], m[2][1], m[3], bool(m[5])) == (r[2], r[3], r[4], r[5], bool(r[6]))
            return dict(
                replay_rows=np.array([r[:7] for r in rows], dtype=np.int32),   # tick_next, ship, iaction, px, py, reward, done
                replay_head_next=np.array([r[7] for r in rows], dtype=np.float64),
                replay_chosen=np.array(chosen, dtype=np.int32),
                replay_maxlen=np.int32(ql.TRAINER.memory.maxlen), replay_len=np.int32(len(mem)))

        run_trace(name, ships, seed, 3, map_every=50, box=box, setup=setup, clock=clock, extra=extra)
        del ql.TRAINER.remember


def epsilon_cases():
    """Exploration schedules (lib/epsilon.py:36-86): value sequences incl. the period wrap and set()."""
    with contextlib.redirect_stdout(_sink):
        from ofighters.lib.epsilon import Epsilon_cos, Epsilon_decay
        out = {}
        e = Epsilon_cos(period=110 * 400)                    # the reference's TRAINER setting, qlearnIA_V2.py:308
        out["cos_44000_first"] = np.array([e.get()] + [e.next() for _ in range(600)])
        e = Epsilon_cos(period=50)
        out["cos_50"] = np.array([e.get()] + [e.next() for _ in range(130)])
        e.set(0.25)
        out["cos_50_after_set"] = np.array([e.t, e.get()] + [e.next() for _ in range(5)])
        d = Epsilon_decay()
        out["decay"] = np.array([d.get()] + [d.next() for _ in range(3000)])
        d.set(0.0105)
        out["decay_after_set"] = np.array([d.next() for _ in range(800)])
    np.savez_compressed(os.path.join(OUT, "epsilon.npz"), **out)
    print("epsilon:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    random_traces()
    brawl_traces()
    crafted_traces()
    raster_cases()
    geometry_cases()
    scratch_nn_cases()
    epsilon_cases()
    replay_traces()
