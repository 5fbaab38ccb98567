"""ctypes binding of libofx.so (include/ofx.h).

The product path: there is NO fallback.  If the shared library is missing the
import raises; if no HIP device is visible every compute call raises
OfxError(OFX_ERR_NO_DEVICE).  Errors surface as plain ``Exception`` subclasses
carrying the library's message, mirroring the reference's bare
``raise Exception(msg)`` convention (lib/battleground.py:30, lib/action.py:40).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libofx.so")

OFX_OK = 0
OFX_ERR_INVALID = -1
OFX_ERR_NO_DEVICE = -2
OFX_ERR_HIP = -3
OFX_ERR_STATE = -4
OFX_ERR_OVERFLOW = -5

BOT_IDLE, BOT_RANDOM, BOT_TURRET, BOT_RUNNER, BOT_THRUST, BOT_SHOOT = range(6)
# agents/agent.py:38-51 behaviour strings
BEHAVIOURS = {None: BOT_IDLE, "idle": BOT_IDLE, "random": BOT_RANDOM, "turret": BOT_TURRET,
              "runner": BOT_RUNNER, "thrust": BOT_THRUST, "shoot": BOT_SHOOT}

(F_SHIP_X, F_SHIP_Y, F_SHIP_PX, F_SHIP_PY, F_SHIP_ALIVE, F_REWARD, F_SCORE, F_N_LASERS, F_LASER_X, F_LASER_Y,
 F_LASER_OWNER, F_LASER_DEAD, F_KILLER, F_TIME, F_LAST_SCORES, F_HULL, F_LASER_DX, F_LASER_DY,
 F_OBS_REWARD) = range(19)

MAP_U8, MAP_F32, MAP_F64, MAP_BITS = range(4)
OPT_TRUNK_PLAIN, OPT_FRAMES_REF, OPT_BILINEAR_LEGACY, OPT_TRUNK_FUSE, OPT_POLICY_BF16, OPT_FIT_PLAIN, OPT_TRUNK_SPARSE = 1, 2, 3, 4, 5, 6, 7   # ofx_set_option


class OfxError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class OfxConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_arenas", "n_ships", "laser_cap", "width", "height", "ship_radius", "laser_radius",
        "ship_speed", "laser_speed", "reward_death", "reward_kill", "reward_aim",
        "reward_trajectory", "episode_ticks", "device", "arena_base")]


class OfxAction(C.Structure):
    _fields_ = [("px", C.c_int32), ("py", C.c_int32), ("shoot", C.c_uint8), ("thrust", C.c_uint8),
                ("valid", C.c_uint8), ("_pad", C.c_uint8)]


class OfxTransition(C.Structure):
    """include/ofx.h: struct ofx_transition - one row of Trainer.memory (qlearnIA_V2.py:237-238)."""
    _fields_ = [("tick_prev", C.c_int32), ("tick_next", C.c_int32), ("frame_prev", C.c_int32),
                ("frame_next", C.c_int32), ("ship", C.c_int32), ("iaction", C.c_int32),
                ("px", C.c_int32), ("py", C.c_int32), ("reward", C.c_int32), ("done", C.c_int32),
                ("head_prev", C.c_float * 8), ("head_next", C.c_float * 8)]


class OfxTensorDesc(C.Structure):
    """include/ofx.h: struct ofx_tensor_desc - a handle-owned HBM array described for a zero-copy view."""
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("itemsize", C.c_int32), ("ndim", C.c_int32),
                ("device", C.c_int32), ("shape", C.c_int64 * 4), ("stride", C.c_int64 * 4)]


DT_U8, DT_I16, DT_I32, DT_I64, DT_F32, DT_F64 = range(6)
DT_TYPESTR = {DT_U8: "|u1", DT_I16: "<i2", DT_I32: "<i4", DT_I64: "<i8", DT_F32: "<f4", DT_F64: "<f8"}


class OfxPolicyDesc(C.Structure):
    _fields_ = [("n_floats", C.c_int32), ("offset", C.c_int32 * 64), ("count", C.c_int32 * 64),
                ("n_tensors", C.c_int32)]


# every symbol include/ofx.h declares: (restype, argtypes)
_vp, _sz, _i, _u64, _u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
SIGNATURES = {
    "ofx_version": (_i, []),
    "ofx_last_error": (C.c_char_p, []),
    "ofx_device_count": (_i, []),
    "ofx_default_config": (None, [C.POINTER(OfxConfig)]),
    "ofx_malloc": (_i, [C.POINTER(_vp), _sz]),
    "ofx_free": (_i, [_vp]),
    "ofx_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "ofx_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "ofx_create": (_i, [C.POINTER(OfxConfig), C.POINTER(_vp)]),
    "ofx_destroy": (_i, [_vp]),
    "ofx_sync": (_i, [_vp]),
    "ofx_stream": (_vp, [_vp]),
    "ofx_set_stream": (_i, [_vp, _vp]),
    "ofx_spawn": (_i, [_vp, _vp]),
    "ofx_restart": (_i, [_vp, _vp, _vp]),
    "ofx_spawn_random": (_i, [_vp, _u64]),
    "ofx_restart_random": (_i, [_vp, _u64, _u32]),
    "ofx_step": (_i, [_vp, _vp]),
    "ofx_rasterise": (_i, [_vp, _i, _vp, _vp]),
    "ofx_map_bytes": (_sz, [_vp, _i]),
    "ofx_map_ptr": (_vp, [_vp, _i, _i]),
    "ofx_observe_head": (_i, [_vp, _vp, _vp]),
    "ofx_bot_actions": (_i, [_vp, _vp, _u64, _u32, _vp]),
    "ofx_rollout": (_i, [_vp, _vp, _u64, _u32, C.c_int32, C.c_int32]),
    "ofx_get_host": (_i, [_vp, _i, _vp, _sz]),
    "ofx_device_ptr": (_vp, [_vp, _i]),
    "ofx_field_bytes": (_sz, [_vp, _i]),
    "ofx_field_desc": (_i, [_vp, _i, C.POINTER(OfxTensorDesc)]),
    "ofx_map_desc": (_i, [_vp, _i, _i, C.POINTER(OfxTensorDesc)]),
    "ofx_overflow_count": (_i, [_vp, C.POINTER(C.c_int64)]),
    "ofx_episode_scores": (_i, [_vp, _vp]),
    "ofx_scores_allreduce": (_i, [_vp, _vp, _vp]),
    "ofx_scratch_feed": (_i, [_vp, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "ofx_scratch_feed_obs": (_i, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "ofx_policy_layout": (_i, [_vp, C.POINTER(OfxPolicyDesc)]),
    "ofx_policy_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ofx_policy_actions": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ofx_policy_explore": (_i, [_vp, C.c_double, _u64, _u32, C.c_int32, _vp, _vp, _vp]),
    "ofx_policy_forward_obs": (_i, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ofx_dqn_targets": (_i, [_vp, _vp, C.c_int32, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "ofx_dqn_fit": (_i, [_vp, _vp, _vp, _vp, C.c_int32, C.c_float, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ofx_dqn_fit_reference": (_i, [_vp, _vp, _vp, _vp, C.c_int32, C.c_float, C.c_int32, _vp, _vp, _vp, C.c_float, _vp, _vp]),
    "ofx_replay_create": (_i, [_vp, C.c_int32, C.c_int32]),
    "ofx_replay_destroy": (_i, [_vp]),
    "ofx_replay_capture": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "ofx_agents_first_done": (_i, [_vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "ofx_replay_count": (_i, [_vp, _vp, _vp]),
    "ofx_replay_rows_host": (_i, [_vp, C.c_int32, _vp, _vp]),
    "ofx_replay_frame_host": (_i, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "ofx_replay_sample": (_i, [_vp, _u64, _u32, C.c_int32, _vp, _vp]),
    "ofx_replay_gather": (_i, [_vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "ofx_replay_gather_valid": (_i, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "ofx_timer_start": (_i, [_vp]),
    "ofx_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "ofx_event_record": (_i, [_vp, C.c_int32]),
    "ofx_policy_profile": (_i, [_vp, C.c_int32]),
    "ofx_policy_pin_weights": (_i, [_vp, _vp]),
    "ofx_set_option": (_i, [_vp, C.c_int32, C.c_int32]),
    "ofx_policy_trunk_stats": (_i, [_vp, C.POINTER(C.c_int64)]),
    "ofx_event_elapsed": (_i, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
}

_LIB = None


def lib():
    """Load libofx.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "ofighters_amd: %s is missing - build it with `make -C ofighters_amd/csrc` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header/library drift
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != OFX_OK:
        raise OfxError(rc, lib().ofx_last_error().decode("utf-8", "replace"))


def default_config(**kw):
    cfg = OfxConfig()
    lib().ofx_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise Exception("unknown config field %r" % k)
        setattr(cfg, k, int(v))
    return cfg
