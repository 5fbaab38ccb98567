"""BASELINE configs[4] at its size on ONE card: 32768 arenas x 8 ships, arena-sharded 8 ways.

Arenas never interact (lib/battleground.py:13-106: all state is per Battleground) and the only cross-shard quantity is
the score every agent banks at Agent.reset (agents/agent.py:61-63), so the 8-GPU run is 8 independent slices of 4096
arenas with global ids r*4096 .. r*4096+4095 plus one [M+1] int64 sum.  Here the 8 slices run one after the other on
the one card of the test box and are compared with ONE unsharded batch of 32768 arenas:
  - the summed episode scores (what the RCCL all-reduce delivers) are equal EXACTLY,
  - every slice's final state equals the matching rows of the unsharded batch bit for bit (ships, lasers, maps),
  - sampled arenas of the LAST slice (arena_base 28672) equal the CPU oracle replaying the same actions.
The RCCL leg itself (8 ranks over xGMI) cannot run on a one-GPU box: unmeasured on hardware, see DESIGN.md 5(e)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHARDS, PER, M, SEED, TICKS, EP = 8, 4096, 8, 0x0F160001, 230, 200
FIELDS = ("F_SHIP_X", "F_SHIP_Y", "F_SHIP_PX", "F_SHIP_PY", "F_SHIP_ALIVE", "F_REWARD", "F_SCORE", "F_LAST_SCORES",
          "F_N_LASERS", "F_LASER_X", "F_LASER_Y", "F_LASER_OWNER", "F_LASER_DEAD", "F_TIME")


def _run(n, base, sample=()):
    """TICKS lock-steps of random-bot step + u8 observation with one episode end (restart at tick EP); the oracle
    replays the sampled local arenas with the actions the device drew."""
    from ofighters_amd import ArenaBatch, _native as nat
    from oracle import pyoracle
    b = ArenaBatch(n, M, arena_base=base)
    b.spawn_random(SEED)
    oracles = {}
    for g in sample:
        o = pyoracle.Arena(n_ships=M)
        o.spawn(pyoracle.reset_draws(o.cfg, SEED, base + int(g), 0))
        oracles[int(g)] = o
    scores = None
    for t in range(TICKS):
        if t == EP:
            b.restart_random(SEED)
            scores = b.episode_scores_host().copy()
            for g, o in oracles.items():
                o.restart(pyoracle.reset_draws(o.cfg, SEED, base + g, b.episode))
        b.bot_actions(["random"] * M, SEED, tick=t)
        if oracles:
            acts = b.actions_host()
            for g, o in oracles.items():
                a = acts[g]
                o.step(np.stack([a["valid"], a["shoot"], a["thrust"], a["px"], a["py"]], axis=1).astype(np.int32))
        b.step()
        b.rasterise()
    state = {f: b.get(getattr(nat, f)).copy() for f in FIELDS}
    if oracles:
        sm, lm = b.maps_host(nat.MAP_U8)
        for g, o in oracles.items():
            s, l = o.ships(), o.lasers()
            k = len(l["x"])
            assert state["F_N_LASERS"][g] == k, (base, g)
            assert np.array_equal(state["F_LASER_X"][g, :k], l["x"]) and np.array_equal(state["F_LASER_Y"][g, :k], l["y"])
            assert np.array_equal(np.stack([state["F_SHIP_X"][g], state["F_SHIP_Y"][g]], 1), s["xy"])
            assert np.array_equal(state["F_SHIP_ALIVE"][g], s["alive"]) and np.array_equal(state["F_REWARD"][g], s["reward"])
            assert np.array_equal(state["F_SCORE"][g], s["score"])
            osm, olm = o.rasterise()
            assert np.array_equal(sm[g], osm) and np.array_equal(lm[g], olm), (base, g)
    # the observation of EVERY arena as packed 1-bit maps (exact and complete: 40 KB per arena instead of 320 KB)
    digest = b.maps_host(nat.MAP_BITS)
    assert b.overflow_count() == 0
    b.close()
    return state, scores, digest


def test_config5_eight_slices_equal_one_unsharded_batch():
    rs = np.random.RandomState(5)
    slices = []
    for r in range(SHARDS):
        sample = np.sort(rs.choice(PER, 12, replace=False)) if r == SHARDS - 1 else ()
        slices.append(_run(PER, r * PER, sample))
    whole, whole_scores, whole_digest = _run(SHARDS * PER, 0)
    # what the all-reduce delivers: the sum of the 8 local [M+1] vectors == the unsharded batch's own sums, exactly
    summed = np.sum([s[1] for s in slices], axis=0)
    assert summed.dtype == np.int64 and np.array_equal(summed, whole_scores)
    assert whole_scores[M] == SHARDS * PER
    for r, (st, _, dg) in enumerate(slices):
        lo, hi = r * PER, (r + 1) * PER
        for f in FIELDS:
            assert np.array_equal(st[f], whole[f][lo:hi]), (r, f)
        for which in (0, 1):
            assert np.array_equal(dg[which], whole_digest[which][lo:hi]), (r, which)
