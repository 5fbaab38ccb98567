"""Diagnostic (GPU box, under rocprofv3 --kernel-trace --stats): DeviceTrainer.replay at one fit batch size.
usage: python tools/fit_profile.py [fit_batch]"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import ArenaBatch
from ofighters_amd.trainer import DeviceTrainer
from ofighters_amd.agents.policy_weights import synthetic
fb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, M = 256, 8
b = ArenaBatch(N, M)
tr = DeviceTrainer(b, synthetic(), batch_size=8, memory_size=32, frames=48, fit_batch=fb)
b.spawn_random(3)
for t in range(12):
    b.bot_actions(["random"] * M, 3, tick=t)
    b.policy_forward(tr.weights.ptr, None); b.policy_explore(0.5, 3, tick=t)
    b.replay_capture(t); b.policy_actions(); b.step()
tr.replay(); b.sync()
t0 = time.perf_counter(); k = 5
for _ in range(k): tr.replay()
b.sync()
print("fit_batch %d: %.2f ms per replay" % (fb, (time.perf_counter() - t0) / k * 1e3))
