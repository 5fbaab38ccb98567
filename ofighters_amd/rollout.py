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
                 to_tensor=None, start_tick=0, probe=None, use_rollout=False):
        self.e = engine
        self.behaviours = list(behaviours)
        self.seed = seed
        self.episode_ticks = episode_ticks
        self.dist = dist
        self.observe = observe
        self.policy = policy          # callable(engine) -> None: overwrite the policy ships' actions
        self.to_tensor = to_tensor    # numpy [M+1] int64 -> tensor usable by dist.all_reduce
        self.probe = probe            # callable(stage, begin[, n_ticks]) around "step" / "obs" (bench.py) or None
        self.use_rollout = use_rollout  # without a policy: K lock-steps per host call through engine.rollout
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
        if self.policy is None and self.use_rollout and hasattr(self.e, "rollout"):
            return self._run_headless(ticks)
        for _ in range(ticks):
            self.lockstep()
        return self.score_log

    def _run_headless(self, ticks):
        """Scripted bots only: whole runs of lock-steps up to the next episode end go down in ONE host call
        (ofx_rollout, the batched Battleground.run of battleground.py:169-173); the restart + score all-reduce at the
        episode ends are the same code as in lockstep()."""
        from . import _native as nat
        left = ticks
        while left > 0:
            if self.tick > 0 and self.tick % self.episode_ticks == 0:
                self._episode_end()
            n = min(left, self.episode_ticks - self.tick % self.episode_ticks)
            p = self.probe
            stage = "obs" if self.observe else "step"
            if p: p(stage, True, n)     # n lock-steps go down in this one call
            self.e.rollout(self.behaviours, self.seed, self.tick, n, nat.MAP_U8 if self.observe else None)
            if p: p(stage, False, n)
            self.tick += n
            left -= n
        return self.score_log


class TrainingRollout(ShardedRollout):
    """The learning agent as a whole: `QlearnIA.play` + `Trainer` (agents/qlearnIA_V2.py:372-418, 199-287) for every
    policy ship of every arena of the shard, in lock-step, all on the device.

    One lock-step =
        scripted bots for the other ships                                      agent.py:99-155
        forward on the trainer's (pinned) weights -> epsilon-greedy / the      :397, :199-235
          collecting phase's random play (first `collecting_steps` steps)      :393-395
        transition capture (remember previous_obs ... obs, done latch)         :388-391, :401-403
        action packing, step, rasterise                                        :447-454, battleground.py:153-166
        epsilon decay (once per lock-step: "all bots share the same trainer")  :398-400
        `trainer.replay(batch_size)` every `replay_every` total steps          :414-415  (total_steps % 50 == 0)
          and on the lock-steps where a learning agent first sees its death    :376-378
        a snapshot every `snapshot_every` episodes                             :416-417

    The reference calls replay once per dying agent; the batched form does at most ONE replay per lock-step (its
    minibatch already draws from every arena's memory), on `DeviceTrainer.fit_batch` rows of what it sampled.  After a fit the trainer's blob has changed in place and
    libofx re-prepares the pinned copy itself (ofx_dqn_fit -> ofx_policy_weights_updated), so the next forward plays
    with the new weights like Keras' shared model does.

    engine    ArenaBatch
    trainer   DeviceTrainer built on that engine (owns weights, Adam state, epsilon, the replay memory)
    policy_ships  ship slots driven by the policy (the stock line-up has ONE, lib/ofighters.py:53); the others
              follow `behaviours`
    """

    def __init__(self, engine, trainer, behaviours, seed, policy_ships=(0,), is_learning=True, collecting_steps=20,
                 replay_every=50, replay_on_death=True, snapshot_every=50, snapshot_folder=None, **kw):
        super().__init__(engine, behaviours, seed, **kw)
        self.trainer = trainer
        self.is_learning = bool(is_learning)
        self.collecting_steps = collecting_steps
        self.replay_every = replay_every
        self.replay_on_death = bool(replay_on_death)
        self.snapshot_every = snapshot_every
        self.snapshot_folder = snapshot_folder     # None: no files are written
        self.total_steps = 0                       # Agent.total_steps of the learning agents (agent.py:68)
        self.episode = 0
        self.losses = []                           # QlearnIA.losses: history['loss'][0] = mse(output1) + mse(output2)
        self.epsilons = []                         # QlearnIA.epsilons: one per episode (:364-365)
        self.snapshots = []
        from .engine import DeviceBuffer
        mk = np.zeros((engine.N, engine.M), np.uint8)
        mk[:, list(policy_ships)] = 1
        self.policy_mask = mk
        engine.sync()
        self._mask = DeviceBuffer(mk.nbytes).upload(mk)
        self._zeros = np.zeros((engine.N, engine.M), np.uint8)
        self._seen_done = DeviceBuffer(mk.nbytes).upload(self._zeros)   # the agents' `done` latches, on the device
        engine.policy_pin_weights(trainer.weights.ptr)
        self.policy = self._play
        self.capture_tick = 0                      # ofx_replay_capture's clock: never restarts at episode ends

    def _episode_end(self):
        total = super()._episode_end()
        self.episode += 1
        self.epsilons.append(self.trainer.epsilon.get())
        self.e.sync()
        self._seen_done.upload(self._zeros)         # QlearnIA.reset: done = False (:360-368)
        if (self.is_learning and self.snapshot_folder is not None and self.snapshot_every
                and self.episode % self.snapshot_every == 0):
            self.snapshots.append(self.trainer.save(id="iteration-%s" % self.episode, overwrite=True,
                                                    folder=self.snapshot_folder))
        return total

    def _replay(self):
        loss = self.trainer.replay()
        if loss is not None:
            self.losses.append(float(loss[0]) + float(loss[1]))
        return loss

    def _play(self, e):
        """QlearnIA.play for every policy ship (the device keeps each agent's done latch and previous_*)."""
        self.total_steps += 1                      # Agent.step increments before bot_play (agent.py:67-68)
        t, m = self.trainer, self._mask.ptr
        collecting = self.total_steps < self.collecting_steps
        replayed = False
        if self.is_learning and self.replay_on_death:
            # obs.done seen for the first time this lock-step (:375-378): counted on the device, one int comes back
            if e.agents_first_done(m, self._seen_done) > 0:
                self._replay()
                replayed = True
        e.policy_forward(t.weights.ptr, m)
        e.policy_explore(t.epsilon.get(), self.seed, tick=self.capture_tick, collecting=collecting, ship_mask_ptr=m)
        e.replay_capture(self.capture_tick, ship_mask_ptr=m)
        e.policy_actions(ship_mask_ptr=m)
        self.capture_tick += 1
        if not collecting and self.is_learning:
            t.decay_epsilon()
        # at most ONE replay per lock-step: the batched replay already draws from every arena's memory, a second one on
        # the same lock-step (deaths AND total_steps % 50 == 0) would fit the same memories twice
        if self.is_learning and self.replay_every and self.total_steps % self.replay_every == 0 and not replayed:
            self._replay()
