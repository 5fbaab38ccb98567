"""Diagnostic: where does the GPU heat map differ from the CPU restatement?  (uses oracle/ as the checker)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from ofighters_amd import ArenaBatch, _native as nat
from oracle import pyoracle

N, M = 3, 3
b = ArenaBatch(N, M)
b.spawn_random(12)
for t in range(35):
    b.bot_actions(["turret"] * (M // 2) + ["random"] * (M - M // 2), 12, tick=t)
    b.step(actions_ptr=b._actions.ptr)
w, _ = pyoracle.policy_init(3, trained_like=False)
out = b.policy_forward_host(w, want_heat=True)
head, _ = b.observe_head()
sm, lm = b.maps_host(nat.MAP_U8)
for g in range(1):
    for i in range(M):
        act, heat, ia, ip = pyoracle.policy_forward(sm[g], lm[g], head[g, i].astype(np.float32), w)
        bad = np.abs(out["heat"][g, i] - heat) > 2e-4 * np.abs(heat).max()
        print("ship", i, "bad", int(bad.sum()))
        ys, xs = np.nonzero(bad)
        if len(ys) == 0: continue
        t = bad.reshape(5, 80, 5, 80).sum(axis=(1, 3))
        print(" per tile (rows = tile row):\n", t)
        ty, tx = np.unravel_index(np.argmax(t), t.shape)
        sub = bad[80 * ty:80 * ty + 80, 80 * tx:80 * tx + 80]
        print(" worst tile", ty, tx, "rows with errors:", np.nonzero(sub.any(axis=1))[0].tolist())
        print(" cols with errors:", np.nonzero(sub.any(axis=0))[0].tolist())
