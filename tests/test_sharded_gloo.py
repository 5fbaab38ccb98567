"""The N>1 path on CPU: two gloo ranks each drive a shard through the same
ShardedRollout host code the GPU bench uses; the all-reduced episodic scores
must equal a single-process run over all arenas (exact: integers)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ofighters_amd.rollout import ShardedRollout, shard_range
from tests.oracle_batch import OracleBatch

SEED, M, PER_RANK, EP, TICKS = 77, 4, 3, 25, 80
BEH = ["turret", "random", "random", "runner"]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, n = shard_range(rank, world, PER_RANK)
    r = ShardedRollout(OracleBatch(n, M, arena_base=base), BEH, SEED, episode_ticks=EP, dist=dist, observe=False,
                       to_tensor=lambda a: torch.from_numpy(a.copy()))
    log = r.run(TICKS)
    if rank == 0:
        np.save(out, np.stack(log))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range():
    assert shard_range(0, 8, 4096) == (0, 4096) and shard_range(7, 8, 4096) == (7 * 4096, 4096)
    with pytest.raises(Exception):
        shard_range(8, 8, 4096)


def test_two_rank_scores_match_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scores.npy")
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        got = np.load(out)
    single = ShardedRollout(OracleBatch(world * PER_RANK, M, arena_base=0), BEH, SEED, episode_ticks=EP, observe=False)
    want = np.stack(single.run(TICKS))
    assert got.shape == want.shape == ((TICKS - 1) // EP, M + 1)
    assert np.array_equal(got, want)
    assert np.all(got[:, M] == world * PER_RANK)       # arena count all-reduced too
    assert got[:, :M].sum() > 0                        # something was scored
