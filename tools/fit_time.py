"""Diagnostic (GPU box): one DeviceTrainer.replay (sample + gather + the two target forwards + the fit) per fit batch size.
usage: python tools/fit_time.py [--plain] [--reference] [fit_batch ...]     (default 8 64 256 1024 4096)
--plain: the layer-by-layer form of the fit (OFX_OPT_FIT_PLAIN; 61 MB of workspace per row: keep it at <= 1024 rows)
--reference: Trainer.replay as written (dense targets) instead of the textbook step"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import ArenaBatch, _native as nat
from ofighters_amd.trainer import DeviceTrainer
from ofighters_amd.agents.policy_weights import synthetic

args = sys.argv[1:]
plain, quirks = "--plain" in args, "--reference" in args
sizes = [int(a) for a in args if not a.startswith("--")] or [8, 64, 256, 1024, 4096]
M = 8
N = max(64, (max(sizes) + 7) // 8)
b = ArenaBatch(N, M)
b.set_option(nat.OPT_FIT_PLAIN, int(plain))
w = synthetic()
tr = DeviceTrainer(b, w, batch_size=8, memory_size=64, frames=24, fit_batch=sizes[0], reference_quirks=quirks)
b.spawn_random(3)
for t in range(12):
    b.bot_actions(["random"] * M, 3, tick=t)
    b.policy_forward(tr.weights.ptr, None); b.policy_explore(0.5, 3, tick=t)
    b.replay_capture(t); b.policy_actions(); b.step(actions_ptr=b._actions.ptr)
for fb in sizes:
    tr.fit_batch = fb
    tr.replay(); b.sync()          # sizes the workspace
    t0 = time.perf_counter(); k = 3
    for _ in range(k): tr.replay()
    b.sync(); dt = (time.perf_counter() - t0) / k
    print("%s%s fit_batch %4d: %8.2f ms per Trainer.replay (%.1f us per row), losses %s"
          % ("plain" if plain else "lean", " reference" if quirks else "", fb, dt * 1e3, dt * 1e6 / fb, tr.losses[-1]), flush=True)
b.close()
