"""ArenaBatch - host-side driver of N lock-stepped arenas on one MI355X.

The batched analogue of ``Battleground`` (lib/battleground.py:10-173): the
per-tick order is the reference's ``frame()`` (battleground.py:163-166)

    request_actions -> generate_frame(actions) -> Observation(battleground)

with the GUI's laser clean-up between ticks (lib/ofighters.py:619-625,702-707)
and ``restart()`` between episodes.  All arithmetic happens in libofx.so
(hand-written HIP); this module only marshals numpy / device pointers.
"""
import ctypes as C

import numpy as np

from . import _native as nat

_FIELD_DTYPE = {
    nat.F_SHIP_X: np.int32, nat.F_SHIP_Y: np.int32, nat.F_SHIP_PX: np.int32, nat.F_SHIP_PY: np.int32,
    nat.F_SHIP_ALIVE: np.uint8, nat.F_REWARD: np.int32, nat.F_SCORE: np.int32, nat.F_N_LASERS: np.int32,
    nat.F_LASER_X: np.float64, nat.F_LASER_Y: np.float64, nat.F_LASER_OWNER: np.uint8,
    nat.F_LASER_DEAD: np.uint8, nat.F_KILLER: np.int16, nat.F_TIME: np.int32, nat.F_LAST_SCORES: np.int32,
    nat.F_HULL: np.int32, nat.F_LASER_DX: np.float64, nat.F_LASER_DY: np.float64, nat.F_OBS_REWARD: np.int32,
}
_MAP_DTYPE = {nat.MAP_U8: np.uint8, nat.MAP_F32: np.float32, nat.MAP_F64: np.float64, nat.MAP_BITS: np.uint8}

ACTION_DTYPE = np.dtype([("px", np.int32), ("py", np.int32), ("shoot", np.uint8), ("thrust", np.uint8),
                         ("valid", np.uint8), ("_pad", np.uint8)])
assert ACTION_DTYPE.itemsize == C.sizeof(nat.OfxAction) == 12


class DeviceBuffer:
    """A raw HBM allocation owned by the host side (ofx_malloc / ofx_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        nat.check(nat.lib().ofx_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise Exception("DeviceBuffer.upload: %d bytes into a %d byte buffer" % (arr.nbytes, self.nbytes))
        nat.check(nat.lib().ofx_memcpy_h2d(self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, dtype, shape, offset=0):
        """Copy `shape` elements of `dtype` starting `offset` bytes into the buffer back to the host."""
        out = np.empty(shape, dtype=dtype)
        if offset < 0 or offset + out.nbytes > self.nbytes:
            raise Exception("DeviceBuffer.download: %d bytes at offset %d from a %d byte buffer" % (out.nbytes, offset, self.nbytes))
        nat.check(nat.lib().ofx_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr + int(offset), out.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            nat.lib().ofx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ArrayInterface:
    """__cuda_array_interface__ (version 3) over a raw HBM pointer; `owner` keeps the memory's owner alive."""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def pack_actions(valid, shoot, thrust, px, py):
    """numpy -> [N][M] ofx_action records (lib/action.py:12-56; valid=0 is None)."""
    valid = np.asarray(valid)
    a = np.zeros(valid.shape, dtype=ACTION_DTYPE)
    a["valid"] = valid
    a["shoot"] = shoot
    a["thrust"] = thrust
    a["px"] = px
    a["py"] = py
    return a


class ArenaBatch:
    def __init__(self, n_arenas, n_ships=8, **cfg):
        self.cfg = nat.default_config(n_arenas=n_arenas, n_ships=n_ships, **cfg)
        h = C.c_void_p()
        nat.check(nat.lib().ofx_create(C.byref(self.cfg), C.byref(h)))
        self._h = h.value
        self.N, self.M, self.L = self.cfg.n_arenas, self.cfg.n_ships, self.cfg.laser_cap
        self.W, self.H = self.cfg.width, self.cfg.height
        self._actions = DeviceBuffer(self.N * self.M * ACTION_DTYPE.itemsize)
        self._draws = None
        self._head = None
        self._done = None
        self.tick = 0
        self.episode = 0

    # ---------------------------------------------------------------- lifetime
    def close(self):
        if getattr(self, "_h", None):
            nat.lib().ofx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def sync(self):
        nat.check(nat.lib().ofx_sync(self._h))

    # --------------------------------------------------------- spawn / restart
    def _draw_buf(self, draws):
        d = np.ascontiguousarray(draws, dtype=np.int32).reshape(self.N, self.M, 2)
        if self._draws is None:
            self._draws = DeviceBuffer(d.nbytes)
        self.sync()  # raw copies are not ordered with the handle's stream
        return self._draws.upload(d).ptr

    def spawn(self, draws):
        """Battleground.__init__ positions: draws [N,M,2] = randint(0,W), randint(0,H)."""
        nat.check(nat.lib().ofx_spawn(self._h, self._draw_buf(draws)))
        self.tick = 0

    def spawn_random(self, seed):
        nat.check(nat.lib().ofx_spawn_random(self._h, seed))
        self.tick = 0

    def restart(self, draws, arena_mask=None):
        mptr = None
        if arena_mask is not None:
            m = np.ascontiguousarray(arena_mask, dtype=np.uint8).reshape(self.N)
            self.sync()
            self._mask = DeviceBuffer(m.nbytes).upload(m)
            mptr = self._mask.ptr
        nat.check(nat.lib().ofx_restart(self._h, self._draw_buf(draws), mptr))
        self.episode += 1

    def restart_random(self, seed):
        self.episode += 1
        nat.check(nat.lib().ofx_restart_random(self._h, seed, self.episode))

    def set_ships(self, x=None, y=None, px=None, py=None):
        """Crafted starts (test / fixture use): overwrite ship fields [N,M]."""
        for f, v in ((nat.F_SHIP_X, x), (nat.F_SHIP_Y, y), (nat.F_SHIP_PX, px), (nat.F_SHIP_PY, py)):
            if v is not None:
                a = np.ascontiguousarray(v, dtype=np.int32).reshape(self.N, self.M)
                self.sync()
                nat.check(nat.lib().ofx_memcpy_h2d(nat.lib().ofx_device_ptr(self._h, f),
                                                   a.ctypes.data_as(C.c_void_p), a.nbytes))

    # -------------------------------------------------------------------- tick
    def step(self, actions=None, actions_ptr=None):
        """One lock-step.  `actions`: structured array [N,M] (pack_actions) or
        `actions_ptr`: device pointer to [N][M] ofx_action."""
        if actions is None and actions_ptr is None:
            actions_ptr = self._actions.ptr  # the device buffer bot_actions / policy_actions wrote
        if actions_ptr is None:
            a = np.ascontiguousarray(actions, dtype=ACTION_DTYPE).reshape(self.N, self.M)
            self.sync()  # the previous tick may still be reading the buffer
            actions_ptr = self._actions.upload(a).ptr
        nat.check(nat.lib().ofx_step(self._h, actions_ptr))
        self.tick += 1

    def bot_actions(self, behaviours, seed, tick=None, out_ptr=None):
        """Scripted bots on device (agents/agent.py:99-155 laws) -> device actions."""
        b = np.array([nat.BEHAVIOURS[x] if not isinstance(x, (int, np.integer)) else int(x) for x in behaviours],
                     dtype=np.int32)
        if b.shape != (self.M,):
            raise Exception("behaviours must have one entry per ship")
        out_ptr = out_ptr or self._actions.ptr
        nat.check(nat.lib().ofx_bot_actions(self._h, b.ctypes.data_as(C.c_void_p), seed,
                                            self.tick if tick is None else tick, out_ptr))
        return out_ptr

    def rollout(self, behaviours, seed, tick0, n_ticks, observe=None):
        """Battleground.run for scripted bots (battleground.py:169-173): n_ticks lock-steps enqueued by ONE host call
        (ofx_rollout); `observe` = a nat.MAP_* type to rasterise after every lock-step, None = step only (all lock-steps
        inside one kernel launch).  Bit-identical to n_ticks x (bot_actions, step[, rasterise])."""
        b = np.array([nat.BEHAVIOURS[x] if not isinstance(x, (int, np.integer)) else int(x) for x in behaviours],
                     dtype=np.int32)
        if b.shape != (self.M,):
            raise Exception("behaviours must have one entry per ship")
        nat.check(nat.lib().ofx_rollout(self._h, b.ctypes.data_as(C.c_void_p), seed, int(tick0), int(n_ticks),
                                        -1 if observe is None else int(observe)))
        self.tick += int(n_ticks)

    def actions_host(self):
        self.sync()
        return self._actions.download(ACTION_DTYPE, (self.N, self.M))

    # ------------------------------------------------------------- observation
    def rasterise(self, map_type=nat.MAP_U8, ship_ptr=None, laser_ptr=None):
        nat.check(nat.lib().ofx_rasterise(self._h, map_type, ship_ptr, laser_ptr))

    def maps_host(self, map_type=nat.MAP_U8):
        """(ship_map, laser_map) as numpy [N,W,H] ([N,W*H/8] for MAP_BITS)."""
        self.rasterise(map_type)
        per = nat.lib().ofx_map_bytes(self._h, map_type)
        dt = np.dtype(_MAP_DTYPE[map_type])
        shape = (self.N, per) if map_type == nat.MAP_BITS else (self.N, self.W, self.H)
        out = []
        for which in (0, 1):
            a = np.empty(shape, dtype=dt)
            self.sync()
            nat.check(nat.lib().ofx_memcpy_d2h(a.ctypes.data_as(C.c_void_p),
                                               nat.lib().ofx_map_ptr(self._h, map_type, which), a.nbytes))
            out.append(a)
        return out

    def observe_head(self):
        """obs.vector[:8] for every ship + done flags (observation.py:101-123)."""
        if self._head is None:
            self._head = DeviceBuffer(self.N * self.M * 8 * 8)
            self._done = DeviceBuffer(self.N * self.M)
        nat.check(nat.lib().ofx_observe_head(self._h, self._head.ptr, self._done.ptr))
        self.sync()
        return (self._head.download(np.float64, (self.N, self.M, 8)),
                self._done.download(np.uint8, (self.N, self.M)))

    # ------------------------------------------------------------ state access
    def get(self, field):
        dt = np.dtype(_FIELD_DTYPE[field])
        nbytes = nat.lib().ofx_field_bytes(self._h, field)
        a = np.empty(nbytes // dt.itemsize, dtype=dt)
        nat.check(nat.lib().ofx_get_host(self._h, field, a.ctypes.data_as(C.c_void_p), nbytes))
        return a if field in (nat.F_N_LASERS, nat.F_TIME) else a.reshape(self.N, -1)

    def device_ptr(self, field):
        return nat.lib().ofx_device_ptr(self._h, field)

    # ------------------------------------------------------------ zero-copy views
    # The batched form of the reference's plugin seam ("any object with .play(obs)", agents/agent.py:34-37): an external
    # policy - a torch module, say - reads the maps and the ships' state where they lie in HBM and writes its actions
    # into the [N][M] ofx_action array the next step() reads.  Nothing is copied; see include/ofx.h (ofx_field_desc) for
    # the ownership rule: the CONTENTS are those of the last step / rasterise and the next one overwrites them in place.
    # torch must have been imported BEFORE the library was loaded (both bring a HIP runtime of the same soname and the
    # process keeps the first one: with libofx's first, torch finds no GPU).
    def _view(self, desc, keep=None):
        import torch
        if not torch.cuda.is_available():
            raise Exception("ArenaBatch tensor views need torch's GPU runtime: import torch before ofighters_amd loads "
                            "libofx.so (the process keeps the first HIP runtime it loads)")
        shape = tuple(int(desc.shape[i]) for i in range(desc.ndim))
        holder = _ArrayInterface(desc.data, shape, nat.DT_TYPESTR[desc.dtype], keep or self)
        return torch.as_tensor(holder, device=torch.device("cuda", int(desc.device)))

    def tensor(self, field):
        """State field `field` (nat.F_*) as a torch tensor over the handle's own memory: [N][M], [N][L] or [N]."""
        d = nat.OfxTensorDesc()
        nat.check(nat.lib().ofx_field_desc(self._h, int(field), C.byref(d)))
        return self._view(d)

    def maps_tensor(self, map_type=nat.MAP_U8):
        """(ship_map, laser_map) of the handle's internal buffers of `map_type` as torch tensors [N][W = y][H = x]
        ([N][W * H / 8] uint8 for MAP_BITS), filled by rasterise(map_type)."""
        out = []
        for which in (0, 1):
            d = nat.OfxTensorDesc()
            nat.check(nat.lib().ofx_map_desc(self._h, int(map_type), which, C.byref(d)))
            out.append(self._view(d))
        return tuple(out)

    def actions_tensor(self):
        """The [N][M] ofx_action array step() reads by default, as views of its fields: px, py int32 [N][M]; shoot,
        thrust, valid uint8 [N][M] (lib/action.py:12-56; valid = 0 is the reference's None)."""
        import torch
        d = nat.OfxTensorDesc()
        d.data, d.dtype, d.itemsize, d.ndim, d.device = self._actions.ptr, nat.DT_U8, 1, 3, self.cfg.device
        d.shape[0], d.shape[1], d.shape[2] = self.N, self.M, ACTION_DTYPE.itemsize
        raw = self._view(d, keep=self._actions)
        words = raw.view(torch.int32)                    # [N][M][3]: px, py, (shoot | thrust << 8 | valid << 16)
        return {"px": words[..., 0], "py": words[..., 1], "shoot": raw[..., 8], "thrust": raw[..., 9],
                "valid": raw[..., 10], "raw": raw}

    def torch_stream(self):
        """The handle's HIP stream as a torch stream: `with torch.cuda.stream(batch.torch_stream()):` puts a policy's
        kernels in order with step() / rasterise() (the handle's stream does not synchronise with torch's default one)."""
        import torch
        return torch.cuda.ExternalStream(int(nat.lib().ofx_stream(self._h)), device=torch.device("cuda", self.cfg.device))

    def overflow_count(self):
        v = C.c_int64(0)
        nat.check(nat.lib().ofx_overflow_count(self._h, C.byref(v)))
        return v.value

    def episode_scores_host(self):
        """int64 [M+1]: per-slot sums banked at the last restart + arena count."""
        buf = DeviceBuffer(8 * (self.M + 1))
        nat.check(nat.lib().ofx_episode_scores(self._h, buf.ptr))
        self.sync()
        return buf.download(np.int64, (self.M + 1,))

    episode_scores = episode_scores_host

    def episode_scores_into(self, dev_ptr):
        nat.check(nat.lib().ofx_episode_scores(self._h, dev_ptr))

    # ------------------------------------------------------------------ timing
    def timer_start(self):
        nat.check(nat.lib().ofx_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        nat.check(nat.lib().ofx_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def event_record(self, idx):
        nat.check(nat.lib().ofx_event_record(self._h, idx))

    def event_elapsed(self, a, b):
        ms = C.c_float(0)
        nat.check(nat.lib().ofx_event_elapsed(self._h, a, b, C.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ policy
    def policy_layout(self):
        d = nat.OfxPolicyDesc()
        nat.check(nat.lib().ofx_policy_layout(self._h, C.byref(d)))
        n = d.n_tensors
        return list(d.offset[:n]), list(d.count[:n]), d.n_floats

    def policy_forward(self, weights_ptr, ship_mask_ptr=None, act_ptr=None, iaction_ptr=None, ipointer_ptr=None,
                       heat_ptr=None):
        """Bi-head forward for every (arena, ship) on the CURRENT state (device pointers;
        None outputs stay inside the handle's workspace for policy_actions)."""
        nat.check(nat.lib().ofx_policy_forward(self._h, weights_ptr, ship_mask_ptr, act_ptr, iaction_ptr,
                                               ipointer_ptr, heat_ptr))

    def policy_pin_weights(self, weights_ptr):
        """Prepare (fold BatchNorm, build the phase weights) once: every following forward on this blob reuses it.
        None unpins.  (The reference keeps one compiled Keras model between predicts, qlearnIA_V2.py:308.)"""
        nat.check(nat.lib().ofx_policy_pin_weights(self._h, weights_ptr))

    def set_option(self, option, value):
        """Diagnostic switches of the forward: nat.OPT_TRUNK_PLAIN, nat.OPT_FRAMES_REF (reference variants)."""
        nat.check(nat.lib().ofx_set_option(self._h, int(option), int(value)))

    def policy_trunk_stats(self):
        """OPT_TRUNK_SPARSE: (M-tiles run, M-tiles, table passes run, table passes) since the last call."""
        v = (C.c_int64 * 4)()
        nat.check(nat.lib().ofx_policy_trunk_stats(self._h, v))
        return tuple(int(x) for x in v)

    def policy_actions(self, out_ptr=None, iaction_ptr=None, ipointer_ptr=None, ship_mask_ptr=None):
        """QlearnIA.play packing of the last forward into [N][M] ofx_action."""
        out_ptr = out_ptr or self._actions.ptr
        nat.check(nat.lib().ofx_policy_actions(self._h, iaction_ptr, ipointer_ptr, ship_mask_ptr, out_ptr))
        return out_ptr

    def policy_forward_host(self, weights, ship_mask=None, want_heat=False):
        """Convenience for tests / the facade: numpy in, numpy out."""
        S = self.N * self.M
        w = np.ascontiguousarray(weights, np.float32)
        dw = DeviceBuffer(w.nbytes).upload(w)
        dm = None
        if ship_mask is not None:
            m = np.ascontiguousarray(ship_mask, np.uint8).reshape(S)
            dm = DeviceBuffer(m.nbytes).upload(m)
        da, di, dp = DeviceBuffer(8 * S), DeviceBuffer(4 * S), DeviceBuffer(8 * S)
        dh = DeviceBuffer(4 * S * self.W * self.H) if want_heat else None
        self.sync()
        self.policy_forward(dw.ptr, dm.ptr if dm else None, da.ptr, di.ptr, dp.ptr, dh.ptr if dh else None)
        self.sync()
        out = dict(act=da.download(np.float32, (self.N, self.M, 2)), iaction=di.download(np.int32, (self.N, self.M)),
                   ipointer=dp.download(np.int32, (self.N, self.M, 2)))
        if want_heat:
            out["heat"] = dh.download(np.float32, (self.N, self.M, self.W, self.H))
        return out

    # ------------------------------------------------------------------ facade support
    def step_packed(self, packed):
        """packed int32 [N,M,5] = valid, shoot, thrust, px, py (the facade's Action.packed())."""
        p = np.asarray(packed, np.int32).reshape(self.N, self.M, 5)
        self.step(pack_actions(p[..., 0], p[..., 1], p[..., 2], p[..., 3], p[..., 4]))

    def snapshot(self):
        g = self.get
        return dict(x=g(nat.F_SHIP_X), y=g(nat.F_SHIP_Y), px=g(nat.F_SHIP_PX), py=g(nat.F_SHIP_PY),
                    alive=g(nat.F_SHIP_ALIVE), hull=g(nat.F_HULL), reward=g(nat.F_REWARD), score=g(nat.F_SCORE),
                    n_lasers=g(nat.F_N_LASERS), lx=g(nat.F_LASER_X), ly=g(nat.F_LASER_Y),
                    lowner=g(nat.F_LASER_OWNER), ldead=g(nat.F_LASER_DEAD))

    def maps_f64(self):
        sm, lm = self.maps_host(nat.MAP_U8)
        return sm.astype(np.float64), lm.astype(np.float64)

    def policy_profile(self, event_base):
        nat.check(nat.lib().ofx_policy_profile(self._h, event_base))

    # ------------------------------------------------------------ replay memory
    TRANSITION_DTYPE = np.dtype([("tick_prev", np.int32), ("tick_next", np.int32), ("frame_prev", np.int32),
                                 ("frame_next", np.int32), ("ship", np.int32),
                                 ("iaction", np.int32), ("px", np.int32), ("py", np.int32), ("reward", np.int32),
                                 ("done", np.int32), ("head_prev", np.float32, 8), ("head_next", np.float32, 8)])

    def replay_create(self, capacity=400, frames=0):
        """Trainer.memory = deque(maxlen=memory_size) per arena (qlearnIA_V2.py:58)."""
        nat.check(nat.lib().ofx_replay_create(self._h, int(capacity), int(frames)))
        self.replay_capacity = int(capacity)

    def replay_capture(self, tick, ship_mask_ptr=None, iaction_ptr=None, ipointer_ptr=None):
        """QlearnIA.play bookkeeping + Trainer.remember for this lock-step; call before step()."""
        nat.check(nat.lib().ofx_replay_capture(self._h, int(tick), ship_mask_ptr, iaction_ptr, ipointer_ptr))

    def agents_first_done(self, ship_mask_ptr, seen_buf):
        """How many selected ships are destroyed now and not yet marked in `seen_buf` (DeviceBuffer, uint8 [N][M]); marks
        them.  QlearnIA.play's `if obs.done` (qlearnIA_V2.py:376-384) for all learning agents: one int crosses to the host."""
        n = C.c_int32(0)
        nat.check(nat.lib().ofx_agents_first_done(self._h, ship_mask_ptr, seen_buf.ptr, C.byref(n)))
        return n.value

    def replay_count(self):
        cnt = np.empty(self.N, np.int32)
        app = np.empty(self.N, np.int64)
        nat.check(nat.lib().ofx_replay_count(self._h, cnt.ctypes.data_as(C.c_void_p), app.ctypes.data_as(C.c_void_p)))
        return cnt, app

    def replay_rows(self, arena):
        """list(memory) of one arena, oldest first, as a structured array."""
        rows = np.zeros(self.replay_capacity, self.TRANSITION_DTYPE)
        n = C.c_int32()
        nat.check(nat.lib().ofx_replay_rows_host(self._h, int(arena), rows.ctypes.data_as(C.c_void_p), C.byref(n)))
        return rows[:n.value]

    def replay_frame(self, arena, tick):
        """(ship_map, laser_map) uint8 [H][W] of a stored lock-step."""
        nb = self.W * self.H // 8
        a, b = np.empty(nb, np.uint8), np.empty(nb, np.uint8)
        nat.check(nat.lib().ofx_replay_frame_host(self._h, int(arena), int(tick), a.ctypes.data_as(C.c_void_p),
                                                   b.ctypes.data_as(C.c_void_p)))
        un = lambda x: np.unpackbits(x, bitorder="little").reshape(self.H, self.W)
        return un(a), un(b)

    def replay_sample(self, seed, draw, batch, slot=None, n=None):
        """random.sample(memory, min(batch, len)) per arena -> (slot DeviceBuffer [N][batch], n DeviceBuffer [N]); the
        caller may hand in both buffers (DeviceTrainer keeps them between replays)."""
        if slot is None:
            slot = DeviceBuffer(4 * self.N * batch)
        if n is None:
            n = DeviceBuffer(4 * self.N)
        nat.check(nat.lib().ofx_replay_sample(self._h, seed, int(draw), int(batch), slot.ptr, n.ptr))
        return slot, n

    def replay_gather(self, slot, batch, with_maps=True):
        """Materialise sampled rows: (rows [N][batch], bits_prev, bits_next uint32 [N][batch][2][W*H/32]) on the host."""
        rows = DeviceBuffer(self.N * batch * self.TRANSITION_DTYPE.itemsize)
        words = self.W * self.H // 32
        bp = DeviceBuffer(4 * self.N * batch * 2 * words) if with_maps else None
        bn = DeviceBuffer(4 * self.N * batch * 2 * words) if with_maps else None
        nat.check(nat.lib().ofx_replay_gather(self._h, slot.ptr, int(batch), rows.ptr, bp.ptr if bp else None,
                                               bn.ptr if bn else None))
        self.sync()
        out = [rows.download(self.TRANSITION_DTYPE, (self.N, batch))]
        if with_maps:
            out += [bp.download(np.uint32, (self.N, batch, 2, words)), bn.download(np.uint32, (self.N, batch, 2, words))]
        return out

    def replay_gather_device(self, slot, batch):
        """Like replay_gather but the minibatch stays in HBM: (rows, bits_prev, bits_next) DeviceBuffers."""
        rows = DeviceBuffer(self.N * batch * self.TRANSITION_DTYPE.itemsize)
        words = self.W * self.H // 32
        bp, bn = DeviceBuffer(4 * self.N * batch * 2 * words), DeviceBuffer(4 * self.N * batch * 2 * words)
        nat.check(nat.lib().ofx_replay_gather(self._h, slot.ptr, int(batch), rows.ptr, bp.ptr, bn.ptr))
        return rows, bp, bn

    def replay_gather_valid(self, slot, n_sampled, batch, first, max_rows):
        """The sampled minibatch without padding rows, packed in (arena, j) order and kept in HBM: entries first ..
        first + max_rows - 1 -> (rows, bits_prev, bits_next, n_rows).  Trainer.replay never pads (qlearnIA_V2.py:241-243)."""
        rows = DeviceBuffer(max_rows * self.TRANSITION_DTYPE.itemsize)
        words = self.W * self.H // 32
        bp, bn = DeviceBuffer(4 * max_rows * 2 * words), DeviceBuffer(4 * max_rows * 2 * words)
        n = C.c_int32()
        nat.check(nat.lib().ofx_replay_gather_valid(self._h, slot.ptr, n_sampled.ptr, int(batch), int(first), int(max_rows),
                                                     rows.ptr, bp.ptr, bn.ptr, C.byref(n)))
        return rows, bp, bn, n.value

    def replay_gather_valid_into(self, slot, n_sampled, batch, first, max_rows, rows, bits_prev, bits_next):
        """replay_gather_valid into the caller's DeviceBuffers (a trainer keeps them between replays); returns n_rows."""
        n = C.c_int32()
        nat.check(nat.lib().ofx_replay_gather_valid(self._h, slot.ptr, n_sampled.ptr, int(batch), int(first), int(max_rows),
                                                     rows.ptr, bits_prev.ptr, bits_next.ptr, C.byref(n)))
        return n.value

    def policy_forward_obs(self, weights_ptr, n_obs, bits_ptr, vec8_ptr, want_probe_ptr=None):
        """Forward on stored observations (Trainer.replay's predictions): host dict of act / iaction / ipointer /
        ptr_max (+ ptr_probe when want_probe_ptr, an int32 [n_obs][2] device array of (x, y), is given)."""
        n = int(n_obs)
        act, ia, ip, pm = DeviceBuffer(8 * n), DeviceBuffer(4 * n), DeviceBuffer(8 * n), DeviceBuffer(4 * n)
        pp = DeviceBuffer(4 * n) if want_probe_ptr else None
        nat.check(nat.lib().ofx_policy_forward_obs(self._h, weights_ptr, n, bits_ptr, vec8_ptr, act.ptr, ia.ptr, ip.ptr,
                                                    pm.ptr, want_probe_ptr, pp.ptr if pp else None))
        self.sync()
        out = {"act": act.download(np.float32, (n, 2)), "iaction": ia.download(np.int32, (n,)),
               "ipointer": ip.download(np.int32, (n, 2)), "ptr_max": pm.download(np.float32, (n,))}
        if pp:
            out["ptr_probe"] = pp.download(np.float32, (n,))
        return out

    def dqn_targets(self, weights_ptr, n, rows_ptr, bits_prev_ptr, bits_next_ptr, gamma=0.9, current=True):
        """Trainer.replay's targets (qlearnIA_V2.py:251-270; gamma = 0.9, :51): host arrays q_sa, p_sp, y_act, y_ptr.
        current=False: only y_act, y_ptr (the forward on `state` is skipped; q_sa / p_sp come back as None)."""
        n = int(n)
        bufs = [DeviceBuffer(4 * n) if (current or k >= 2) else None for k in range(4)]
        nat.check(nat.lib().ofx_dqn_targets(self._h, weights_ptr, n, rows_ptr, bits_prev_ptr, bits_next_ptr, float(gamma),
                                             *[b.ptr if b else None for b in bufs]))
        self.sync()
        return tuple(b.download(np.float32, (n,)) if b else None for b in bufs)

    def dqn_fit(self, weights_buf, adam_m_buf, adam_v_buf, step, lr, n, rows_ptr, bits_prev_ptr, y_act_ptr, y_ptr_ptr,
                grad_buf=None):
        """One model.fit step of Trainer.replay on device (qlearnIA_V2.py:284); DeviceBuffers are updated in place.
        Returns (mse(output1), mse(output2))."""
        loss = (C.c_float * 2)()
        nat.check(nat.lib().ofx_dqn_fit(self._h, weights_buf.ptr, adam_m_buf.ptr, adam_v_buf.ptr, int(step), float(lr),
                                         int(n), rows_ptr, bits_prev_ptr, y_act_ptr, y_ptr_ptr,
                                         grad_buf.ptr if grad_buf else None, loss))
        return float(loss[0]), float(loss[1])

    def dqn_fit_reference(self, weights_buf, adam_m_buf, adam_v_buf, step, lr, n, rows_ptr, bits_prev_ptr, bits_next_ptr,
                          gamma=0.9, grad_buf=None):
        """The fit step with Trainer.replay's quirks as written (qlearnIA_V2.py:251-285: whole-prediction targets,
        ptr_target[x][y], inputs = next_state); computes its own targets.  Returns (mse(output1), mse(output2))."""
        loss = (C.c_float * 2)()
        nat.check(nat.lib().ofx_dqn_fit_reference(self._h, weights_buf.ptr, adam_m_buf.ptr, adam_v_buf.ptr, int(step),
                                                   float(lr), int(n), rows_ptr, bits_prev_ptr, bits_next_ptr, float(gamma),
                                                   grad_buf.ptr if grad_buf else None, loss))
        return float(loss[0]), float(loss[1])

    def policy_explore(self, epsilon, seed, tick=None, collecting=False, ship_mask_ptr=None, iaction_ptr=None,
                       ipointer_ptr=None):
        """epsilon-greedy / collecting-phase random play over the last forward's results."""
        nat.check(nat.lib().ofx_policy_explore(self._h, float(epsilon), seed, self.tick if tick is None else tick,
                                               int(bool(collecting)), ship_mask_ptr, iaction_ptr, ipointer_ptr))
