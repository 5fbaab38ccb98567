"""Soak run (GPU box): TrainingRollout on 1024 arenas for 450 lock-steps with 1024-row fits on the reference's replay schedule -
losses and weights stay finite, the TD loss falls (r03: 281 replays in 10 s, loss 5.5 -> 0.17)."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import ArenaBatch
from ofighters_amd.lib.epsilon import Epsilon_decay
from ofighters_amd.rollout import TrainingRollout
from ofighters_amd.trainer import DeviceTrainer
from ofighters_amd.agents.policy_weights import synthetic
N, M = 1024, 8
b = ArenaBatch(N, M)
eps = Epsilon_decay(); eps.set(0.3)
tr = DeviceTrainer(b, synthetic(), epsilon=eps, batch_size=8, memory_size=64, frames=96, fit_batch=1024, learning_rate=1e-4)
roll = TrainingRollout(b, tr, ["random"] * M, 0x0F160001, policy_ships=(0,), episode_ticks=b.cfg.episode_ticks)
t0 = time.perf_counter()
roll.run(450)
b.sync()
L = np.array(roll.losses)
print("lock-steps 450, replays", len(L), "seconds %.1f" % (time.perf_counter() - t0))
print("finite:", bool(np.isfinite(L).all()), " first 5 mean", L[:5].mean(0), " last 5 mean", L[-5:].mean(0))
w = tr.weights_host()
print("weights finite:", bool(np.isfinite(w).all()), "episodes", b.episode, "scores", roll.score_log[-1] if roll.score_log else None)
b.close()
