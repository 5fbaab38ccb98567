import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ofighters_amd import ArenaBatch
from ofighters_amd.lib.epsilon import Epsilon_decay
from ofighters_amd.rollout import TrainingRollout
from ofighters_amd.trainer import DeviceTrainer
from ofighters_amd.agents.policy_weights import synthetic
N, M = 4096, 8
b = ArenaBatch(N, M)
eps = Epsilon_decay(); eps.set(0.1)
tr = DeviceTrainer(b, synthetic(), epsilon=eps, batch_size=8, memory_size=64, frames=96, fit_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 256)
roll = TrainingRollout(b, tr, ["random"] * M, 1, policy_ships=(0,))
roll.run(30); b.sync()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); roll.run(30); b.sync(); dt = time.perf_counter() - t0
pr.disable()
print("ms per tick", dt / 30 * 1e3, "replays", len(roll.losses))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
