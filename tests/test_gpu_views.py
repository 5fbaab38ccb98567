"""Zero-copy views of the handle's arrays (include/ofx.h: ofx_field_desc / ofx_map_desc; ArenaBatch.tensor / maps_tensor /
actions_tensor) and an EXTERNAL batched policy on them - the batch form of the reference's plugin seam, "any object with
.play(obs)" (agents/agent.py:34-37; Trainer.get_best_action reads obs.ship_map / laser_map / vector[:8],
agents/qlearnIA_V2.py:206-220): a small torch policy reads the maps and the ships' state in HBM and writes the ofx_action
fields in place, on the handle's stream.

The GPU leg runs in a child process that imports torch BEFORE libofx.so is loaded: both bring a HIP runtime with the same
soname and a process keeps the first one it loads (in the test session libofx's is already there, and torch then finds
no GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
import torch                      # first: see the module docstring
sys.path.insert(0, %r)
from ofighters_amd import ArenaBatch, _native as nat
from ofighters_amd.engine import pack_actions

N, M, seed, ticks = 48, 8, 0x0F160001, 50
a, b = ArenaBatch(N, M), ArenaBatch(N, M)
for e in (a, b):
    e.spawn_random(seed)
    e.rasterise(nat.MAP_U8)


def policy(ship_map, laser_map, x, y, alive, tick):
    """10 lines of torch: point every ship at the centre of mass of its arena's laser map (the arena centre when it is
    empty), shoot on even (ship + tick), thrust otherwise; destroyed ships answer None (valid = 0)."""
    n = laser_map.shape[0]
    lm = laser_map.to(torch.float32)
    mass = lm.sum(dim=(1, 2)).clamp(min=1.0)
    cols = torch.arange(lm.shape[2], device=lm.device, dtype=torch.float32)
    rows = torch.arange(lm.shape[1], device=lm.device, dtype=torch.float32)
    cx = torch.where(lm.sum(dim=(1, 2)) > 0, (lm.sum(dim=1) * cols).sum(dim=1) / mass, torch.full((n,), 200.0, device=lm.device))
    cy = torch.where(lm.sum(dim=(1, 2)) > 0, (lm.sum(dim=2) * rows).sum(dim=1) / mass, torch.full((n,), 200.0, device=lm.device))
    px = cx.to(torch.int32)[:, None].expand(-1, x.shape[1]) + (ship_map.sum(dim=(1, 2)) %% 7).to(torch.int32)[:, None]
    py = cy.to(torch.int32)[:, None].expand(-1, x.shape[1])
    shoot = ((torch.arange(x.shape[1], device=x.device)[None, :] + tick) %% 2 == 0).expand(n, -1)
    return px, py, shoot.to(torch.uint8), (~shoot).to(torch.uint8), alive.clone()


# the views: same memory as the C-ABI's device pointers, the shapes / dtypes of include/ofx.h
xa, ya, alive_a = a.tensor(nat.F_SHIP_X), a.tensor(nat.F_SHIP_Y), a.tensor(nat.F_SHIP_ALIVE)
assert xa.data_ptr() == a.device_ptr(nat.F_SHIP_X) and tuple(xa.shape) == (N, M) and xa.dtype == torch.int32
assert alive_a.dtype == torch.uint8 and a.tensor(nat.F_LASER_X).dtype == torch.float64
assert tuple(a.tensor(nat.F_LASER_X).shape) == (N, a.L) and tuple(a.tensor(nat.F_TIME).shape) == (N,)
sm_a, lm_a = a.maps_tensor(nat.MAP_U8)
assert sm_a.data_ptr() == nat.lib().ofx_map_ptr(a.handle, nat.MAP_U8, 0) and tuple(sm_a.shape) == (N, 400, 400)
act = a.actions_tensor()
sa = a.torch_stream()
sm_b, lm_b = b.maps_tensor(nat.MAP_U8)
xb, yb, alive_b = b.tensor(nat.F_SHIP_X), b.tensor(nat.F_SHIP_Y), b.tensor(nat.F_SHIP_ALIVE)
sb = b.torch_stream()
for t in range(ticks):
    # A: the policy's outputs go straight into the action array in HBM, on the handle's stream; nothing touches the host
    with torch.cuda.stream(sa):
        px, py, sh, th, va = policy(sm_a, lm_a, xa, ya, alive_a, t)
        act["px"].copy_(px); act["py"].copy_(py); act["shoot"].copy_(sh); act["thrust"].copy_(th); act["valid"].copy_(va)
    a.step()
    a.rasterise(nat.MAP_U8)
    # B: the same policy, its answers through the host and pack_actions (the path the facade takes)
    with torch.cuda.stream(sb):
        px, py, sh, th, va = policy(sm_b, lm_b, xb, yb, alive_b, t)
    sb.synchronize()
    b.step(pack_actions(va.cpu().numpy(), sh.cpu().numpy(), th.cpu().numpy(), px.cpu().numpy(), py.cpu().numpy()))
    b.rasterise(nat.MAP_U8)
a.sync(); b.sync()
for f in (nat.F_SHIP_X, nat.F_SHIP_Y, nat.F_SHIP_PX, nat.F_SHIP_PY, nat.F_SHIP_ALIVE, nat.F_REWARD, nat.F_SCORE, nat.F_N_LASERS,
          nat.F_LASER_X, nat.F_LASER_Y, nat.F_LASER_OWNER, nat.F_LASER_DEAD, nat.F_KILLER, nat.F_TIME, nat.F_HULL):
    ga, gb = a.get(f), b.get(f)
    assert np.array_equal(ga, gb), f
    assert np.array_equal(a.tensor(f).cpu().numpy().reshape(ga.shape), ga), f      # the view IS the state
assert np.array_equal(sm_a.cpu().numpy(), b.maps_host(nat.MAP_U8)[0])
assert int(a.get(nat.F_SHIP_ALIVE).sum()) < N * M and int(a.get(nat.F_N_LASERS).sum()) > 0    # something happened
a.close(); b.close()
print("views ok")
'''


@pytest.mark.gpu
def test_external_torch_policy_on_zero_copy_views():
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "views ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_descriptor_struct_matches_the_header():
    """CPU: the ctypes mirror of ofx_tensor_desc has the header's fields in the header's order and sizes."""
    import ctypes as C
    import re
    from ofighters_amd import _native as nat
    src = open(os.path.join(ROOT, "include", "ofx.h")).read()
    body = re.search(r"typedef struct ofx_tensor_desc \{(.*?)\} ofx_tensor_desc;", src, re.S).group(1)
    names = re.findall(r"(?:void \*|int32_t |int64_t )(\w+)(?:\[4\])?;", body)
    assert names == [f[0] for f in nat.OfxTensorDesc._fields_]
    assert C.sizeof(nat.OfxTensorDesc) == 8 + 4 * 4 + 2 * 4 * 8
    assert [int(v) for v in re.findall(r"OFX_DT_\w+ = (\d)", src)] == [nat.DT_U8, nat.DT_I16, nat.DT_I32, nat.DT_I64, nat.DT_F32, nat.DT_F64]
