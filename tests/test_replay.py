"""Replay memory (transition capture).

CPU: the restatement oracle/pyoracle.ReplayMemory is pinned against every Trainer.remember() call the live
reference made in tests/golden/step_replay_*.npz (QlearnIA ships in their collecting phase, see
oracle/gen_golden.py:replay_traces).
GPU (-m gpu): ofx_replay_* through the C-ABI against those fixtures and against the restatement on seeded random
rollouts; everything is integer / exact (heads are small integers in float32)."""
import numpy as np
import pytest

from oracle import pyoracle
from tests.trace_util import load_trace, unpack_map

TRACES = ["replay_open", "replay_brawl"]


def _chosen_table(z):
    return {(int(t), int(i)): (int(ia), (int(px), int(py))) for t, i, ia, px, py in z["replay_chosen"]}


def _drive_oracle(z):
    """Feed the recorded observations / choices of the reference to the restatement; returns it."""
    M = z["init_state"].shape[0]
    ticks, episodes = int(z["ticks"]), int(z["episodes"])
    ships = sorted({int(i) for i in z["replay_chosen"][:, 1]})
    chosen = _chosen_table(z)
    mem = pyoracle.ReplayMemory(M, capacity=int(z["replay_maxlen"]))
    log = []
    for t in range(ticks * episodes):
        if t and t % ticks == 0:
            mem.reset()                            # Battleground.restart -> Agent.reset
        for i in ships:
            before = mem.appended
            played = mem.play(t, i, z["obs8"][t, i], bool(z["obs_done"][t, i]), chosen.get((t, i), (0, (0, 0))))
            assert played == ((t, i) in chosen), (t, i)      # the done latch: play() chooses nothing once done
            if mem.appended > before:
                log.append(mem.memory[-1])
    return mem, log, ships


@pytest.mark.parametrize("name", TRACES)
def test_replay_oracle_matches_reference(name):
    z = load_trace(name)
    mem, log, _ = _drive_oracle(z)
    ref = z["replay_rows"]                          # tick_next, ship, iaction, px, py, reward, done  (append order)
    assert len(log) == len(ref)
    got = np.array([[r[1], r[2], r[3], r[4], r[5], r[6], r[7]] for r in log], np.int32)
    assert np.array_equal(got, ref)
    assert all(r[0] == r[1] - 1 for r in log)       # play() runs every tick until done: state is the previous tick
    assert np.array_equal(np.stack([r[9] for r in log]), z["replay_head_next"].astype(np.float32))
    assert np.array_equal(np.stack([r[8] for r in log]),
                          np.stack([z["obs8"][r[0], r[2]] for r in log]).astype(np.float32))
    # the deque keeps exactly the last maxlen rows
    assert len(mem.rows()) == int(z["replay_len"]) == min(len(ref), int(z["replay_maxlen"]))
    tail = np.array([[r[1], r[2]] for r in mem.rows()], np.int32)
    assert np.array_equal(tail, ref[-len(tail):, :2])
    assert ref[:, 6].sum() > 0 and len(ref) > int(z["replay_maxlen"])   # the fixture exercises done rows and maxlen


def test_replay_sample_is_a_uniform_subset():
    seen = np.zeros(10, int)
    for draw in range(2000):
        s = pyoracle.replay_sample(10, 4, 0x0F160003, 7, draw)
        assert len(s) == len(set(s)) == 4 and all(0 <= v < 10 for v in s)
        seen[s] += 1
    assert seen.min() > 700 and seen.max() < 900       # 800 expected per index
    assert pyoracle.replay_sample(3, 8, 1, 0, 0) != [] and sorted(pyoracle.replay_sample(3, 8, 1, 0, 0)) == [0, 1, 2]
    assert pyoracle.replay_sample(0, 8, 1, 0, 0) == []


# ------------------------------------------------------------------------------------------------------ GPU
def _rows_equal(dev, ora):
    assert len(dev) == len(ora), (len(dev), len(ora))
    for d, o in zip(dev, ora):
        assert (d["tick_prev"], d["tick_next"], d["ship"], d["iaction"], d["px"], d["py"], d["reward"], d["done"]) == \
               tuple(int(v) for v in o[:8]), (d, o)
        assert np.array_equal(d["head_prev"], o[8]) and np.array_equal(d["head_next"], o[9])


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRACES)
@pytest.mark.parametrize("frames", [0, 64])
def test_replay_capture_trace_gpu(name, frames):
    from ofighters_amd import ArenaBatch, DeviceBuffer, pack_actions, _native as nat
    z = load_trace(name)
    M = z["init_state"].shape[0]
    ticks, episodes = int(z["ticks"]), int(z["episodes"])
    mem, _, ships = _drive_oracle(z)
    chosen = _chosen_table(z)
    REP = 4
    rep = lambda a: np.broadcast_to(a, (REP,) + a.shape)
    b = ArenaBatch(REP, M, laser_cap=256)
    b.replay_create(int(z["replay_maxlen"]), frames)
    b.spawn(rep(z["spawn_draws"]))
    mask = np.zeros((REP, M), np.uint8)
    mask[:, ships] = 1
    mask_d = DeviceBuffer(mask.nbytes).upload(mask)
    ia_d, ip_d = DeviceBuffer(4 * REP * M), DeviceBuffer(8 * REP * M)
    t = 0
    for ep in range(episodes):
        for _ in range(ticks):
            ia = np.full((REP, M), -7, np.int32)          # poison: a latched agent's choice must never be stored
            ip = np.full((REP, M, 2), -7, np.int32)
            for i in ships:
                if (t, i) in chosen:
                    ia[:, i], ip[:, i] = chosen[(t, i)][0], chosen[(t, i)][1]
            b.sync()
            ia_d.upload(ia), ip_d.upload(ip)
            b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
            a = z["actions"][t]
            b.step(pack_actions(rep(a[:, 0]), rep(a[:, 1]), rep(a[:, 2]), rep(a[:, 3]), rep(a[:, 4])))
            t += 1
        b.restart(rep(z["reset_draws"][ep]))
    cnt, app = b.replay_count()
    assert np.all(cnt == len(mem.rows())) and np.all(app == mem.appended)
    for r in (0, REP - 1):
        _rows_equal(b.replay_rows(r), mem.rows())
    # stored frames = the observation maps BEFORE the step of that lock-step = the trace's maps after lock-step t-1
    # an arena stores a frame on every lock-step where one of its agents plays; the ring keeps the last F of them
    T = ticks * episodes
    C = int(z["replay_maxlen"])
    F = frames if frames else C + C // 4 + 2
    stored = sorted({t for (t, _) in chosen})
    ring = set(stored[-F:])
    checked = 0
    for k, mt in enumerate(z["map_ticks"]):
        tick = int(mt) + 1
        if tick >= T or tick % ticks == 0:             # after a restart the trace holds no map of the new episode
            continue
        if tick not in ring:
            with pytest.raises(Exception, match="not in the frame ring"):
                b.replay_frame(0, tick)
            continue
        sm, lm = b.replay_frame(REP - 1, tick)
        assert np.array_equal(sm, unpack_map(z["ship_maps"][k])) and np.array_equal(lm, unpack_map(z["laser_maps"][k]))
        checked += 1
    assert checked >= 1 or frames
    # minibatch: per-arena Floyd sample == restatement; rows whose state frame was overwritten are not eligible
    rows = mem.rows()
    skip = sum(1 for r in rows if r[0] not in ring)
    assert all(r[0] not in ring for r in rows[:skip]) and (skip > 0) == (frames == 64)
    for draw, batch in ((0, 8), (1, 30), (2, 500)):
        slot, n = b.replay_sample(0x0F160003, draw, batch)
        b.sync()
        s = slot.download(np.int32, (REP, batch))
        nn = n.download(np.int32, (REP,))
        for r in range(REP):
            want = [skip + v for v in pyoracle.replay_sample(len(rows) - skip, batch, 0x0F160003, r, draw)]
            assert nn[r] == len(want) and list(s[r, :nn[r]]) == want and np.all(s[r, nn[r]:] == -1)
        out_rows, bp, bn = b.replay_gather(slot, batch)
        for j in range(batch):
            v = int(s[1, j])
            if v < 0:
                assert out_rows[1, j]["ship"] == -1
                continue
            _rows_equal([out_rows[1, j]], [rows[v]])
            for bits, tk in ((bp[1, j], rows[v][0]), (bn[1, j], rows[v][1])):
                sm, lm = b.replay_frame(1, tk)
                got = np.unpackbits(bits.view(np.uint8), bitorder="little").reshape(2, 400, 400)
                assert np.array_equal(got[0], sm) and np.array_equal(got[1], lm)
    b.close()


@pytest.mark.gpu
def test_replay_random_rollout_vs_oracle():
    """Device bots + device exploration (collecting phase) on 64 arenas x 6 ships, 3 capturing ships per arena,
    episodes of 60 lock-steps: rows of sampled arenas equal the restatement fed with the device's own observations."""
    from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
    N, M, seed, cap = 64, 6, 0x0F160001, 50
    b = ArenaBatch(N, M)
    b.replay_create(cap, 0)
    b.spawn_random(seed)
    mask = np.zeros((N, M), np.uint8)
    mask[:, [0, 2, 5]] = 1
    mask_d = DeviceBuffer(mask.nbytes).upload(mask)
    ia_d, ip_d = DeviceBuffer(4 * N * M), DeviceBuffer(8 * N * M)
    mems = {g: pyoracle.ReplayMemory(M, cap) for g in (0, 17, 63)}
    for t in range(150):
        if t and t % 60 == 0:
            b.restart_random(seed)
            for m in mems.values():
                m.reset()
        b.bot_actions(["random"] * M, seed, tick=t)
        b.policy_explore(1.0, seed, tick=t, collecting=True, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr,
                         ipointer_ptr=ip_d.ptr)
        b.policy_actions(out_ptr=b._actions.ptr, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
        b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
        head, done = b.observe_head()
        ia, ip = ia_d.download(np.int32, (N, M)), ip_d.download(np.int32, (N, M, 2))
        for g, m in mems.items():
            for i in (0, 2, 5):
                m.play(t, i, head[g, i], bool(done[g, i]), (ia[g, i], ip[g, i]))
        b.step(actions_ptr=b._actions.ptr)
    cnt, app = b.replay_count()
    for g, m in mems.items():
        assert cnt[g] == len(m.rows()) and app[g] == m.appended
        _rows_equal(b.replay_rows(g), m.rows())
    assert app.max() > cap and any(r[7] for m in mems.values() for r in m.rows())
    b.close()


@pytest.mark.gpu
def test_replay_errors():
    from ofighters_amd import ArenaBatch
    b = ArenaBatch(4, 4)
    b.spawn_random(1)
    with pytest.raises(Exception, match="before ofx_replay_create"):
        b.replay_capture(0)
    with pytest.raises(Exception, match="capacity must be > 0"):
        b.replay_create(0)
    b.close()


@pytest.mark.gpu
def test_dqn_targets_vs_oracle():
    """Replay memory -> sample -> gather -> ofx_dqn_targets (two forwards on the stored observations + the TD
    arithmetic of Trainer.replay, qlearnIA_V2.py:251-270) against the CPU restatement of the policy forward run on
    the unpacked frames.  Also checks ofx_policy_forward_obs on live observations against ofx_policy_forward."""
    from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
    N, M, seed, cap, batch, gamma = 8, 5, 0x0F160001, 40, 3, 0.9
    b = ArenaBatch(N, M)
    b.replay_create(cap, 0)
    b.spawn_random(seed)
    w, _ = pyoracle.policy_init(6, trained_like=True)
    dw = DeviceBuffer(w.nbytes).upload(w)
    mask = np.zeros((N, M), np.uint8)
    mask[:, [1, 3]] = 1
    mask_d = DeviceBuffer(mask.nbytes).upload(mask)
    ia_d, ip_d = DeviceBuffer(4 * N * M), DeviceBuffer(8 * N * M)
    for t in range(45):
        b.bot_actions(["random"] * M, seed, tick=t)
        b.policy_explore(1.0, seed, tick=t, collecting=True, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr,
                         ipointer_ptr=ip_d.ptr)
        b.policy_actions(out_ptr=b._actions.ptr, ship_mask_ptr=mask_d.ptr, iaction_ptr=ia_d.ptr, ipointer_ptr=ip_d.ptr)
        b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
        b.step(actions_ptr=b._actions.ptr)
    slot, _ = b.replay_sample(0x0F160003, 0, batch)
    rows_d, bp_d, bn_d = b.replay_gather_device(slot, batch)
    n = N * batch
    q_sa, p_sp, y_act, y_ptr = b.dqn_targets(dw.ptr, n, rows_d.ptr, bp_d.ptr, bn_d.ptr, gamma)
    # the targets alone (q_sa = p_sp = NULL: no forward on `state`): the same bits
    none_a, none_b, y_act2, y_ptr2 = b.dqn_targets(dw.ptr, n, rows_d.ptr, bp_d.ptr, bn_d.ptr, gamma, current=False)
    assert none_a is None and none_b is None and np.array_equal(y_act, y_act2) and np.array_equal(y_ptr, y_ptr2)
    rows = rows_d.download(b.TRANSITION_DTYPE, (n,))
    bp = bp_d.download(np.uint32, (n, 2, 5000))
    bn = bn_d.download(np.uint32, (n, 2, 5000))
    assert (rows["ship"] >= 0).sum() >= 10
    unpack = lambda x: np.unpackbits(x.view(np.uint8), bitorder="little").reshape(2, 400, 400)
    checked = 0
    for i in range(0, n, 3):
        r = rows[i]
        if r["ship"] < 0:
            assert q_sa[i] == p_sp[i] == y_act[i] == y_ptr[i] == 0
            continue
        m0, m1 = unpack(bp[i]), unpack(bn[i])
        act0, heat0, _, _ = pyoracle.policy_forward(m0[0], m0[1], r["head_prev"], w)
        act1, heat1, _, _ = pyoracle.policy_forward(m1[0], m1[1], r["head_next"], w)
        live = 0.0 if r["done"] else 1.0
        tol_a = 2e-5 * max(1.0, float(np.abs(act0).max()), float(np.abs(act1).max()))
        tol_h = 2e-5 * max(float(np.abs(heat0).max()), float(np.abs(heat1).max()))
        assert abs(q_sa[i] - act0[r["iaction"]]) <= tol_a
        assert abs(p_sp[i] - heat0[r["py"], r["px"]]) <= tol_h
        assert abs(y_act[i] - (r["reward"] + gamma * act1.max() * live)) <= tol_a
        assert abs(y_ptr[i] - (r["reward"] + gamma * heat1.max() * live)) <= tol_h
        checked += 1
    assert checked >= 4
    # forward_obs on the live observation == the arena forward (same kernels, one trunk run per observation)
    full = b.policy_forward_host(w)
    b.rasterise(nat.MAP_BITS)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    pick = [(0, 1), (3, 3), (7, 0)]
    bits = np.stack([np.stack([np.packbits(sm[g].ravel(), bitorder="little"), np.packbits(lm[g].ravel(), bitorder="little")])
                     for g, _ in pick]).view(np.uint32)
    vec = np.stack([head[g, i] for g, i in pick]).astype(np.float32)
    probe = np.array([[full["ipointer"][g, i][0], full["ipointer"][g, i][1]] for g, i in pick], np.int32)
    b.sync()
    bits_d = DeviceBuffer(bits.nbytes).upload(bits)
    vec_d = DeviceBuffer(vec.nbytes).upload(vec)
    probe_d = DeviceBuffer(probe.nbytes).upload(probe)
    o = b.policy_forward_obs(dw.ptr, len(pick), bits_d.ptr, vec_d.ptr, probe_d.ptr)
    for k, (g, i) in enumerate(pick):
        assert np.array_equal(o["act"][k], full["act"][g, i]) and o["iaction"][k] == full["iaction"][g, i]
        assert tuple(o["ipointer"][k]) == tuple(full["ipointer"][g, i])
        assert o["ptr_probe"][k] == o["ptr_max"][k]          # the probe sits on the arg-max
    b.close()


@pytest.mark.gpu
def test_replay_partial_restart_and_small_capacity():
    """ofx_restart with an arena mask resets previous_* / the done latch of the masked arenas only (QlearnIA.reset per
    restarted Battleground); capacity 3 keeps the last 3 rows; M = 64 ships fill a whole wave."""
    from ofighters_amd import ArenaBatch, DeviceBuffer
    N, M, seed = 3, 64, 5
    b = ArenaBatch(N, M)
    b.replay_create(3, 8)
    b.spawn_random(seed)
    mask = np.zeros((N, M), np.uint8)
    mask[:, 0] = 1                                   # one capturing ship per arena
    mask_d = DeviceBuffer(mask.nbytes).upload(mask)
    ia = np.zeros((N, M), np.int32)
    ip = np.zeros((N, M, 2), np.int32)
    ia_d, ip_d = DeviceBuffer(ia.nbytes), DeviceBuffer(ip.nbytes)

    def tick(t):
        ia[:, 0], ip[:, 0, 0], ip[:, 0, 1] = t % 2, 10 + t, 20 + t
        b.sync()
        ia_d.upload(ia), ip_d.upload(ip)
        b.bot_actions(["idle"] * M, seed, tick=t)    # nobody moves or shoots: nobody dies
        b.replay_capture(t, mask_d.ptr, ia_d.ptr, ip_d.ptr)
        b.step(actions_ptr=b._actions.ptr)

    for t in range(3):
        tick(t)
    cnt, app = b.replay_count()
    assert list(app) == [2, 2, 2]                    # transitions 0->1, 1->2
    b.restart(np.full((N, M, 2), 7, np.int32), arena_mask=[0, 1, 0])     # only arena 1 starts a new episode
    for t in range(3, 6):
        tick(t)
    cnt, app = b.replay_count()
    assert list(app) == [5, 4, 5] and list(cnt) == [3, 3, 3]             # arena 1 has no row across its restart
    r0, r1 = b.replay_rows(0), b.replay_rows(1)
    assert [(r["tick_prev"], r["tick_next"]) for r in r0] == [(2, 3), (3, 4), (4, 5)]
    assert [(r["tick_prev"], r["tick_next"]) for r in r1] == [(1, 2), (3, 4), (4, 5)]
    assert [(r["iaction"], r["px"], r["py"]) for r in r0] == [(0, 12, 22), (1, 13, 23), (0, 14, 24)]   # previous_* of tick_prev
    assert all(r["ship"] == 0 and r["done"] == 0 for r in r0)
    b.close()
