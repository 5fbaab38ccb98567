"""CPU-side checks of the C-ABI boundary: the shared library loads, exports
every symbol include/ofx.h declares, and fails loudly (no CPU fallback) when
no HIP device is visible.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from ofighters_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ofx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ofx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(nat.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), "libofx.so does not export %s" % s
    # and the Python binding covers exactly the header
    assert sorted(nat.SIGNATURES) == syms


def test_struct_layouts():
    assert C.sizeof(nat.OfxAction) == 12
    assert C.sizeof(nat.OfxConfig) == 16 * 4
    cfg = nat.default_config()
    # reference constants: observation.py:10-11, ship.py:24,43,86, laser.py:23, qlearnIA_V2.py:39-44, ofighters.py:59
    assert (cfg.width, cfg.height, cfg.ship_radius, cfg.laser_radius) == (400, 400, 8, 2)
    assert (cfg.ship_speed, cfg.laser_speed, cfg.episode_ticks) == (8, 10, 200)
    assert (cfg.reward_death, cfg.reward_kill, cfg.reward_aim, cfg.reward_trajectory) == (0, 0, 2, 1)
    assert nat.lib().ofx_version() == 1


def test_invalid_config_messages():
    h = C.c_void_p()
    for kw in (dict(n_ships=0), dict(n_ships=65), dict(laser_cap=100), dict(width=0), dict(width=401, height=401)):
        cfg = nat.default_config(**kw)
        rc = nat.lib().ofx_create(C.byref(cfg), C.byref(h))
        assert rc == nat.OFX_ERR_INVALID
        assert nat.lib().ofx_last_error()
    with pytest.raises(Exception):
        nat.default_config(bogus=1)


def test_no_cpu_fallback_without_device():
    if nat.lib().ofx_device_count() > 0:
        pytest.skip("a HIP device is visible")
    h = C.c_void_p()
    cfg = nat.default_config()
    rc = nat.lib().ofx_create(C.byref(cfg), C.byref(h))
    assert rc == nat.OFX_ERR_NO_DEVICE
    assert b"no CPU fallback" in nat.lib().ofx_last_error()
    from ofighters_amd import ArenaBatch, OfxError
    with pytest.raises(OfxError):
        ArenaBatch(4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ofighters_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "libofx_oracle" not in txt and "from oracle" not in txt, f
