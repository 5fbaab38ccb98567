import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from ofighters_amd import ArenaBatch, DeviceBuffer
from ofighters_amd.trainer import DeviceTrainer
from ofighters_amd.agents.policy_weights import synthetic
N, M = 64, 8
b = ArenaBatch(N, M); w = synthetic()
for fb in (8, 64):
    tr = DeviceTrainer(b, w, batch_size=8, memory_size=64, frames=24, fit_batch=fb)
    b.spawn_random(3)
    for t in range(12):
        b.bot_actions(["random"] * M, 3, tick=t)
        b.policy_forward(tr.weights.ptr, None); b.policy_explore(0.5, 3, tick=t)
        b.replay_capture(t); b.policy_actions(); b.step(actions_ptr=b._actions.ptr)
    tr.replay(); b.sync()
    t0 = time.perf_counter(); k = 3
    for _ in range(k): tr.replay()
    b.sync(); dt = (time.perf_counter() - t0) / k
    print("fit_batch %d: %.1f ms per Trainer.replay (sample + gather of %d rows + 2 target forwards + fit), losses %s" % (fb, dt * 1e3, N * 8, tr.losses[-1]))
b.close()
