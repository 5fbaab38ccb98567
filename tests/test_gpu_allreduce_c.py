"""ofx_scores_allreduce: the path's one collective for a C consumer of libofx.so - RCCL's
ncclAllReduce(sum, int64) of the [M+1] episode scores on the handle's stream.  A one-GPU box can only form a ONE-rank
communicator (RCCL refuses two ranks per device), so this checks the call sequence a C host makes - ncclGetUniqueId,
ncclCommInitRank, ofx_scores_allreduce - on hardware at world size 1, where the result must equal ofx_episode_scores.
The 8-rank leg over xGMI is unmeasured on hardware.  Runs in a child process without torch, like a C
host: torch ships its own RCCL copy."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
rccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)
from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat

class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]

b = ArenaBatch(512, 8, arena_base=1024)
b.spawn_random(9)
b.rollout(["random"] * 8, 9, 0, 200)
b.restart_random(9)
want = b.episode_scores_host()
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
out = DeviceBuffer(8 * 9)
nat.check(nat.lib().ofx_scores_allreduce(b.handle, comm, out.ptr))
b.sync()
got = out.download(np.int64, (9,))
assert np.array_equal(got, want), (got, want)
assert got[8] == 512 and got[:8].sum() > 0
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
b.close()
print("allreduce ok", got.tolist())
""" % ROOT


def test_scores_allreduce_one_rank_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "allreduce ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
