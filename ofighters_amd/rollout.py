"""Sharded rollout driver: the host logic of the N>1 path.

Arenas never interact (all state is per Battleground; the only cross-arena
object of the reference is the read-only policy singleton,
agents/qlearnIA_V2.py:308,342), so the path shards trivially: rank r of W owns
the contiguous block of global arenas [r*n, (r+1)*n).  The counter RNG is keyed
by the GLOBAL arena id, so any world size reproduces the same arenas.  The one
collective is the all-reduce of the per-slot episode scores banked by
Agent.reset (agents/agent.py:61-63) at each episode end - RCCL over xGMI on the
GPUs (backend "nccl"), gloo in the CPU tests.

The engine is duck-typed (ArenaBatch on the GPU; tests drive the same code with
an oracle-backed stand-in under gloo, world_size 2).
"""
import numpy as np


def shard_range(rank, world, arenas_per_rank):
    """(first global arena id, count) of a rank's shard - weak scaling."""
    if not (0 <= rank < world):
        raise Exception("rank %d outside world of %d" % (rank, world))
    return rank * arenas_per_rank, arenas_per_rank


class ShardedRollout:
    """Lock-step loop of one shard: actions -> step -> observation, episode
    restarts every `episode_ticks` (lib/ofighters.py:59,684-688) and the
    episodic score all-reduce.

    engine   object with spawn_random / restart_random / bot_actions / step /
             rasterise / episode_scores(as int64 numpy [M+1]) and attributes
             N, M, episode (ArenaBatch on the GPU - bench.py drives THIS class at every
             world size - or the oracle-backed stand-in of the gloo test)
    dist     a torch.distributed-like module or None (single process)
    """

    def __init__(self, engine, behaviours, seed, episode_ticks=200, dist=None, observe=True, policy=None,
                 to_tensor=None, start_tick=0, probe=None):
        self.e = engine
        self.behaviours = list(behaviours)
        self.seed = seed
        self.episode_ticks = episode_ticks
        self.dist = dist
        self.observe = observe
        self.policy = policy          # callable(engine) -> None: overwrite the policy ships' actions
        self.to_tensor = to_tensor    # numpy [M+1] int64 -> tensor usable by dist.all_reduce
        self.probe = probe            # callable(stage, begin) around "step" / "obs" (bench.py: HIP events) or None
        self.tick = start_tick        # a start inside an episode puts its end where the caller wants it (bench.py)
        self.score_log = []           # all-reduced [M+1] per finished episode
        self.e.spawn_random(seed)

    def _episode_end(self):
        self.e.restart_random(self.seed)
        local = np.asarray(self.e.episode_scores(), dtype=np.int64)
        if self.dist is not None:
            t = self.to_tensor(local)
            self.dist.all_reduce(t)           # SUM over ranks: the only collective of the path
            total = np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t, dtype=np.int64)
        else:
            total = local
        self.score_log.append(total.copy())
        return total

    def lockstep(self):
        if self.tick > 0 and self.tick % self.episode_ticks == 0:
            self._episode_end()
        self.e.bot_actions(self.behaviours, self.seed, tick=self.tick)
        if self.policy is not None:
            self.policy(self.e)
        p = self.probe
        if p: p("step", True)
        self.e.step()
        if p: p("step", False)
        if self.observe:
            if p: p("obs", True)
            self.e.rasterise()
            if p: p("obs", False)
        self.tick += 1

    def run(self, ticks):
        for _ in range(ticks):
            self.lockstep()
        return self.score_log
