"""ctypes wrapper around oracle/libofx_oracle.so.

TEST INFRASTRUCTURE: may be imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product package (ofighters_amd) never
imports this module and has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_ships", "width", "height", "ship_radius", "laser_radius", "ship_speed",
        "laser_speed", "reward_death", "reward_kill", "reward_aim", "reward_trajectory")]


def default_cfg(n_ships=8, **kw):
    cfg = OrcCfg(n_ships, 400, 400, 8, 2, 8, 10, 0, 0, 2, 1)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def build(force=False):
    so = os.path.join(_HERE, "libofx_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ofx_oracle.c", "policy_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libofx_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32p, u8p, i64p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(OrcCfg)]
        L.orc_destroy.argtypes = [vp]
        L.orc_spawn.argtypes = [vp, i32p]
        L.orc_set_ship.argtypes = [vp, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long]
        L.orc_restart.argtypes = [vp, i32p]
        L.orc_step.argtypes = [vp, i32p]
        L.orc_n_lasers.argtypes = [vp]
        L.orc_n_lasers.restype = C.c_int
        L.orc_get_ships.argtypes = [vp, i32p, i32p, u8p, i64p, i64p, i32p, i64p, i64p]
        L.orc_get_lasers.argtypes = [vp, f64p, f64p, i32p, u8p]
        L.orc_obs_head.argtypes = [vp, f64p, u8p]
        L.orc_rasterise.argtypes = [vp, u8p, u8p]
        L.orc_disk.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_thrust.argtypes = [C.POINTER(OrcCfg), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_long, C.c_long]
        L.orc_edge.argtypes = [C.c_long] * 6 + [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.orc_edge.restype = C.c_int
        L.orc_enemy_aimed.argtypes = [C.POINTER(OrcCfg)] + [C.c_long] * 4
        L.orc_enemy_aimed.restype = C.c_int
        L.orc_enemy_on_trajectory.argtypes = [C.POINTER(OrcCfg)] + [C.c_long] * 6
        L.orc_enemy_on_trajectory.restype = C.c_int
        L.orc_nn_feed.argtypes = [i32p, C.c_int, f64p, f64p, f64p, f64p]
        L.orc_philox.argtypes = [C.c_uint32] * 4 + [C.c_uint64, vp]
        L.orc_bot_actions.argtypes = [vp, i32p, C.c_uint64, C.c_uint32, C.c_uint32, i32p]
        L.orc_reset_draws.argtypes = [C.POINTER(OrcCfg), C.c_uint64, C.c_uint32, C.c_uint32, i32p]
        L.orc_run_random.argtypes = [C.POINTER(OrcCfg), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.orc_run_random.restype = C.c_uint64
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Arena:
    """One reference Battleground restated on the CPU."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if cfg is not None else default_cfg(**kw)
        self.M = self.cfg.n_ships
        self._h = lib().orc_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def spawn(self, draws):
        d = np.ascontiguousarray(draws, dtype=np.int32).reshape(self.M, 2)
        lib().orc_spawn(self._h, _p(d))

    def set_ship(self, i, x, y, px, py):
        lib().orc_set_ship(self._h, i, int(x), int(y), int(px), int(py))

    def restart(self, draws):
        d = np.ascontiguousarray(draws, dtype=np.int32).reshape(self.M, 2)
        lib().orc_restart(self._h, _p(d))

    def step(self, actions):
        """actions [M,5] int32: valid, shoot, thrust, px, py"""
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.M, 5)
        lib().orc_step(self._h, _p(a))

    def ships(self):
        M = self.M
        out = dict(xy=np.zeros((M, 2), np.int32), pt=np.zeros((M, 2), np.int32), alive=np.zeros(M, np.uint8),
                   reward=np.zeros(M, np.int64), score=np.zeros(M, np.int64), killer=np.zeros(M, np.int32),
                   last_score=np.zeros(M, np.int64), hull=np.zeros(M, np.int64))
        lib().orc_get_ships(self._h, _p(out["xy"]), _p(out["pt"]), _p(out["alive"]), _p(out["reward"]),
                            _p(out["score"]), _p(out["killer"]), _p(out["last_score"]), _p(out["hull"]))
        return out

    def lasers(self):
        n = lib().orc_n_lasers(self._h)
        out = dict(x=np.zeros(n), y=np.zeros(n), owner=np.zeros(n, np.int32), destroyed=np.zeros(n, np.uint8))
        if n:
            lib().orc_get_lasers(self._h, _p(out["x"]), _p(out["y"]), _p(out["owner"]), _p(out["destroyed"]))
        return out

    def obs_head(self):
        head = np.zeros((self.M, 8))
        done = np.zeros(self.M, np.uint8)
        lib().orc_obs_head(self._h, _p(head), _p(done))
        return head, done

    def rasterise(self):
        W, H = self.cfg.width, self.cfg.height
        sm = np.zeros((W, H), np.uint8)
        lm = np.zeros((W, H), np.uint8)
        lib().orc_rasterise(self._h, _p(sm), _p(lm))
        return sm, lm

    def bot_actions(self, behaviours, seed, global_arena, tick):
        b = np.ascontiguousarray(behaviours, dtype=np.int32)
        act = np.zeros((self.M, 5), np.int32)
        lib().orc_bot_actions(self._h, _p(b), C.c_uint64(seed), global_arena, tick, _p(act))
        return act


def disk(r, c, radius, rows=400, cols=400):
    m = np.zeros((rows, cols), np.uint8)
    lib().orc_disk(_p(m), rows, cols, float(r), float(c), float(radius))
    return m


def thrust(cfg, x, y, px, py):
    cx, cy = C.c_long(int(x)), C.c_long(int(y))
    lib().orc_thrust(C.byref(cfg), C.byref(cx), C.byref(cy), int(px), int(py))
    return cx.value, cy.value


def edge(x, y, radius, xn, yn, distance):
    ex, ey = C.c_long(0), C.c_long(0)
    ok = lib().orc_edge(int(x), int(y), int(radius), int(xn), int(yn), int(distance), C.byref(ex), C.byref(ey))
    return (ex.value, ey.value) if ok else None


def enemy_aimed(cfg, px, py, ex, ey):
    return bool(lib().orc_enemy_aimed(C.byref(cfg), int(px), int(py), int(ex), int(ey)))


def enemy_on_trajectory(cfg, sx, sy, px, py, ex, ey):
    return bool(lib().orc_enemy_on_trajectory(C.byref(cfg), int(sx), int(sy), int(px), int(py), int(ex), int(ey)))


def nn_feed(layers, weights, biases, x):
    layers = np.ascontiguousarray(layers, dtype=np.int32)
    w = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).ravel() for a in weights]))
    b = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).ravel() for a in biases]))
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(int(layers[-1]))
    lib().orc_nn_feed(_p(layers), len(layers), _p(w), _p(b), _p(x), _p(y))
    return y


def philox(c0, c1, c2, c3, seed):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(c0, c1, c2, c3, C.c_uint64(seed), _p(out))
    return out


def reset_draws(cfg, seed, global_arena, episode):
    d = np.zeros((cfg.n_ships, 2), np.int32)
    lib().orc_reset_draws(C.byref(cfg), C.c_uint64(seed), global_arena, episode, _p(d))
    return d


def run_random(cfg, n_arenas, ticks, seed, do_raster, episode_ticks=200):
    return lib().orc_run_random(C.byref(cfg), n_arenas, ticks, C.c_uint64(seed), int(do_raster), episode_ticks)


def bench_run(cfg, weights, n_arenas, ticks, seed, do_raster, episode_ticks=200, n_pol=0, n_threads=1):
    """bench.py's cpu_baseline legs as one C call: the orc_run_random loop, optionally with n_pol policy forwards per
    arena and lock-step, over n_threads POSIX threads (arena g -> thread g % n_threads)."""
    L = lib()
    L.orc_bench_run.argtypes = [C.POINTER(OrcCfg), C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                C.c_int]
    L.orc_bench_run.restype = C.c_uint64
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    return L.orc_bench_run(C.byref(cfg), None if w is None else _p(w), int(n_arenas), int(ticks), C.c_uint64(seed),
                           int(do_raster), int(episode_ticks), int(n_pol), int(n_threads))


# ------------------------------------------------------------------ policy
def policy_layout():
    """(offsets, counts, total) of the float32 weight blob (oracle/policy_oracle.c header)."""
    L = lib()
    L.orc_policy_layout.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_policy_layout.restype = C.c_int
    off = np.zeros(64, np.int32)
    cnt = np.zeros(64, np.int32)
    n = L.orc_policy_layout(_p(off), _p(cnt))
    return off[:n].copy(), cnt[:n].copy(), int(off[n])


POLICY_TENSORS = (
    [("conv%d.%s" % (i, s)) for i in (1, 2, 3, 4) for s in ("kernel", "bias", "gamma", "beta", "mean", "var")]
    + ["dense1.kernel", "dense1.bias", "dense2.kernel", "dense2.bias", "output1.kernel", "output1.bias",
       "updense1.kernel", "updense1.bias"]
    + [("upconv%d.%s" % (i, s)) for i in (1, 2, 3) for s in ("kernel", "bias", "gamma", "beta", "mean", "var")]
    + ["upconv4.kernel", "upconv4.bias"])


def policy_init(seed, trained_like=False):
    """Synthetic weights of the pointer_model architecture (no checkpoint ships with
    the reference): he_uniform convs, glorot_uniform dense, zero bias, BN gamma=1
    beta=0 mean=0 var=1 as at construction (qlearnIA_V2.py:129-186).  With
    trained_like=True biases and BN statistics are randomised too so every term
    of the graph is exercised."""
    off, cnt, total = policy_layout()
    rs = np.random.RandomState(seed)
    w = np.zeros(total, np.float32)
    shapes = {}
    cin = {"conv1": 2, "conv2": 8, "conv3": 8, "conv4": 8, "upconv1": 1, "upconv2": 2, "upconv3": 4, "upconv4": 8}
    cout = {"conv1": 8, "conv2": 8, "conv3": 8, "conv4": 8, "upconv1": 2, "upconv2": 4, "upconv3": 8, "upconv4": 1}
    dense = {"dense1": (5008, 100), "dense2": (100, 50), "output1": (50, 2), "updense1": (100, 625)}
    for name, o, c in zip(POLICY_TENSORS, off, cnt):
        layer, kind = name.split(".")
        if kind == "kernel":
            if layer in dense:
                fi, fo = dense[layer]
                lim = np.sqrt(6.0 / (fi + fo))
                shapes[name] = (fi, fo)
            else:
                lim = np.sqrt(6.0 / (9 * cin[layer]))
                shapes[name] = (3, 3, cin[layer], cout[layer])
            w[o:o + c] = rs.uniform(-lim, lim, c)
        elif kind == "gamma":
            w[o:o + c] = rs.uniform(0.5, 1.5, c) if trained_like else 1.0
        elif kind == "var":
            w[o:o + c] = rs.uniform(0.5, 2.0, c) if trained_like else 1.0
        elif kind in ("beta", "mean", "bias"):
            w[o:o + c] = rs.uniform(-0.2, 0.2, c) if trained_like else 0.0
        if name not in shapes:
            shapes[name] = (c,)
    return w, {n: (int(o), shapes[n]) for n, o in zip(POLICY_TENSORS, off)}


def policy_forward(ship_map, laser_map, vec8, weights, want_heat=True, legacy_bilinear=False):
    L = lib()
    L.orc_policy_forward2.argtypes = [C.c_void_p] * 8 + [C.c_int]
    sm = np.ascontiguousarray(ship_map, np.uint8)
    lm = np.ascontiguousarray(laser_map, np.uint8)
    v = np.ascontiguousarray(vec8, np.float32)
    w = np.ascontiguousarray(weights, np.float32)
    act = np.zeros(2, np.float32)
    heat = np.zeros((400, 400), np.float32) if want_heat else None
    ia = np.zeros(1, np.int32)
    ip = np.zeros(2, np.int32)
    L.orc_policy_forward2(_p(sm), _p(lm), _p(v), _p(w), _p(act), _p(heat), _p(ia), _p(ip), int(bool(legacy_bilinear)))
    return act, heat, int(ia[0]), (int(ip[0]), int(ip[1]))


def policy_explore(cfg, eps, seed, global_arena, ship, tick, collecting=False):
    L = lib()
    L.orc_policy_explore.argtypes = [C.POINTER(OrcCfg), C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_int, C.c_void_p]
    L.orc_policy_explore.restype = C.c_int
    out = np.zeros(3, np.int32)
    hit = L.orc_policy_explore(C.byref(cfg), float(eps), C.c_uint64(seed), global_arena, ship, tick, int(collecting), _p(out))
    return (int(out[0]), int(out[1]), int(out[2])) if hit else None


# ---- replay memory (test infrastructure; small cases only - pure Python) -------------------------------------------
class ReplayMemory:
    """CPU restatement of Trainer.memory = deque(maxlen=memory_size) (agents/qlearnIA_V2.py:58), Trainer.remember
    (:237-238) and the bookkeeping QlearnIA.play does around the action choice (:370-403) for the capturing ships of
    ONE arena.  Pinned by tests/golden/step_replay_*.npz (every remember() call of the live reference, in order).

    A row is (tick_prev, tick_next, ship, iaction, px, py, reward, done, head_prev, head_next).  The reference stores
    references to Observation objects that every ship of a tick shares and mutates (lib/battleground.py:150), so the
    `vector` found in its memory is the LAST analysed ship's; this restatement (like the device code) keeps the
    capturing ship's own head instead - the value `next_state.vector[:8]` has at the moment remember() is called."""

    def __init__(self, n_ships, capacity=400):
        from collections import deque
        self.memory = deque(maxlen=capacity)
        self.appended = 0
        self.M = n_ships
        self.reset()

    def reset(self, ships=None):
        """QlearnIA.reset (:360-368): done = False, previous_obs = previous_action = previous_pointer = None."""
        if ships is None:
            self.latched = [False] * self.M
            self.prev = [None] * self.M
        else:
            for i in ships:
                self.latched[i], self.prev[i] = False, None

    def play(self, tick, ship, head8, obs_done, chosen):
        """One QlearnIA.play(obs) call.  `chosen` = (iaction, (px, py)) is what the action choice returns when it is
        reached (it is not reached once the agent is done).  Returns True when an action was chosen."""
        if self.latched[ship]:                      # :373-374
            return False
        if obs_done:                                # :376-383 (replay()/fit is out of scope)
            self.latched[ship] = True
        reward = int(head8[0])                      # obs.reward (observation.py:103)
        if self.prev[ship] is not None:             # :385-387
            pt, ph, ia, (px, py) = self.prev[ship]
            self.memory.append((pt, tick, ship, ia, px, py, reward, int(bool(obs_done)),
                                np.asarray(ph, np.float32), np.asarray(head8, np.float32)))
            self.appended += 1
        ia, ip = chosen                             # :389-396
        self.prev[ship] = (tick, np.array(head8, np.float64), int(ia), (int(ip[0]), int(ip[1])))   # :397-399
        return True

    def rows(self):
        return list(self.memory)


def replay_sample(capacity_count, batch, seed, global_arena, draw):
    """random.sample(memory, min(batch, len(memory))) (qlearnIA_V2.py:241-243) on the counter RNG: Floyd's subset
    sampling, draw j from Philox counter (global arena, j, draw, stream 3).  Returns row indices, oldest first."""
    n = min(batch, capacity_count)
    out = []
    for j in range(n):
        top = capacity_count - n + j
        r = philox(global_arena, j, draw, 3, seed)
        t = (int(r[0]) * (top + 1)) >> 32
        if t in out:
            t = top
        out.append(t)
    return out
