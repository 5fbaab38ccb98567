"""GPU parity of the observation rasteriser and the scratch-MLP forward."""
import ctypes as C

import numpy as np
import pytest

from tests.trace_util import GOLDEN, unpack_map

pytestmark = pytest.mark.gpu


def _load(name):
    zf = np.load(GOLDEN + "/" + name)
    return {k: zf[k] for k in zf.files}


def test_raster_cases_gpu():
    """Every golden disc (skimage.draw.disk through Circle.binary_draw) drawn as a
    laser (float centre, r=2) or a ship (int centre, r=8) on the GPU."""
    from ofighters_amd import ArenaBatch, _native as nat
    z = _load("raster_cases.npz")
    cases, maps = z["cases"], z["maps"]
    n = len(cases)
    b = ArenaBatch(n, 1, laser_cap=64)
    b.spawn(np.zeros((n, 1, 2), np.int32))
    is_ship = cases[:, 2] == 8.0
    # ships: put the ship at the centre; lasers: kill the ship, plant one laser
    x = np.where(is_ship, cases[:, 0], 0).astype(np.int32)
    y = np.where(is_ship, cases[:, 1], 0).astype(np.int32)
    b.set_ships(x=x[:, None], y=y[:, None])
    lib = nat.lib()
    lx = np.zeros((n, 64)); ly = np.zeros((n, 64))
    lx[:, 0] = cases[:, 0]; ly[:, 0] = cases[:, 1]
    nl = (~is_ship).astype(np.int32)
    alive = is_ship.astype(np.uint8)
    b.sync()
    for f, a in ((nat.F_LASER_X, lx), (nat.F_LASER_Y, ly), (nat.F_N_LASERS, nl), (nat.F_SHIP_ALIVE, alive)):
        a = np.ascontiguousarray(a)
        nat.check(lib.ofx_memcpy_h2d(lib.ofx_device_ptr(b.handle, f), a.ctypes.data_as(C.c_void_p), a.nbytes))
    for mt in (nat.MAP_U8, nat.MAP_F32, nat.MAP_F64):
        sm, lm = b.maps_host(mt)
        for k in range(n):
            want = unpack_map(maps[k])
            got = sm[k] if is_ship[k] else lm[k]
            other = lm[k] if is_ship[k] else sm[k]
            assert np.array_equal(got, want.astype(got.dtype)), (k, cases[k], mt)
            assert not other.any()
    sb, lb = b.maps_host(nat.MAP_BITS)
    for k in range(n):
        assert np.array_equal(sb[k] if is_ship[k] else lb[k], maps[k])
    b.close()


def test_raster_vs_oracle_rollout():
    """Maps of a live rollout (incl. just-destroyed lasers and dead ships) equal
    the oracle's skimage restatement, for every element type."""
    from ofighters_amd import ArenaBatch, _native as nat
    from oracle import pyoracle
    N, M, seed = 32, 8, 21
    b = ArenaBatch(N, M)
    b.spawn_random(seed)
    oracles = []
    for g in range(N):
        o = pyoracle.Arena(n_ships=M)
        o.spawn(pyoracle.reset_draws(o.cfg, seed, g, 0))
        oracles.append(o)
    for t in range(120):
        b.bot_actions(["turret"] * 3 + ["random"] * 5, seed, tick=t)
        acts = b.actions_host()
        b.step(actions_ptr=b._actions.ptr)
        for g, o in enumerate(oracles):
            a = acts[g]
            o.step(np.stack([a["valid"], a["shoot"], a["thrust"], a["px"], a["py"]], axis=1).astype(np.int32))
        if t % 15 == 14:
            for mt in (nat.MAP_U8, nat.MAP_F32, nat.MAP_F64):
                sm, lm = b.maps_host(mt)
                for g, o in enumerate(oracles):
                    osm, olm = o.rasterise()
                    assert np.array_equal(sm[g], osm.astype(sm.dtype)), (t, g, mt)
                    assert np.array_equal(lm[g], olm.astype(lm.dtype)), (t, g, mt)
            sb, lb = b.maps_host(nat.MAP_BITS)
            for g, o in enumerate(oracles):
                osm, olm = o.rasterise()
                assert np.array_equal(sb[g], np.packbits(osm)) and np.array_equal(lb[g], np.packbits(olm))
    b.close()


def test_raster_full_size_properties():
    """4096 x 8 (BASELINE config 3): pixel counts bounded by the disc areas,
    maps are exactly {0,1}, and a checksum over all arenas equals the sum of the
    oracle-verified per-disc pixel counts for integer-centred ships."""
    from ofighters_amd import ArenaBatch, _native as nat
    from oracle import pyoracle
    N, M = 4096, 8
    b = ArenaBatch(N, M)
    b.spawn_random(7)
    for t in range(30):
        b.bot_actions(["random"] * M, 7, tick=t)
        b.step(actions_ptr=b._actions.ptr)
    sm, lm = b.maps_host(nat.MAP_U8)
    assert sm.max() <= 1 and lm.max() <= 1
    alive = b.get(nat.F_SHIP_ALIVE)
    nl = b.get(nat.F_N_LASERS)
    per_ship = sm.reshape(N, -1).sum(1)
    assert np.all(per_ship <= 193 * alive.sum(1))         # r=8 integer centre = 193 cells
    per_laser = lm.reshape(N, -1).sum(1)
    assert np.all(per_laser <= 16 * nl)                   # r=2: a 5x5 box minus corners bounds any centre
    # exact check on a sample
    x, y = b.get(nat.F_SHIP_X), b.get(nat.F_SHIP_Y)
    lx, ly = b.get(nat.F_LASER_X), b.get(nat.F_LASER_Y)
    for g in np.random.RandomState(1).choice(N, 24, replace=False):
        want_s = np.zeros((400, 400), np.uint8)
        for i in range(M):
            if alive[g, i]:
                want_s |= pyoracle.disk(float(y[g, i]), float(x[g, i]), 8.0)
        want_l = np.zeros((400, 400), np.uint8)
        for j in range(nl[g]):
            want_l |= pyoracle.disk(ly[g, j], lx[g, j], 2.0)
        assert np.array_equal(sm[g], want_s) and np.array_equal(lm[g], want_l)
    b.close()


def test_scratch_nn_gpu():
    """Neural_network.feed golden vectors + the oracle on a bigger random net."""
    from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
    from oracle import pyoracle
    z = _load("scratch_nn.npz")
    b = ArenaBatch(1, 1)
    lib = nat.lib()

    def feed(layers, W, B, X):
        layers = np.ascontiguousarray(layers, np.int32)
        w = np.concatenate([a.ravel() for a in W]); bb = np.concatenate([a.ravel() for a in B])
        X = np.ascontiguousarray(X, np.float64)
        dw, db, dx = DeviceBuffer(w.nbytes).upload(w), DeviceBuffer(bb.nbytes).upload(bb), DeviceBuffer(X.nbytes).upload(X)
        dy, da = DeviceBuffer(8 * len(X) * int(layers[-1])), DeviceBuffer(4 * len(X))
        nat.check(lib.ofx_scratch_feed(b.handle, layers.ctypes.data_as(C.c_void_p), len(layers), dw.ptr, db.ptr,
                                       dx.ptr, len(X), dy.ptr, da.ptr))
        b.sync()
        return dy.download(np.float64, (len(X), int(layers[-1]))), da.download(np.int32, (len(X),))

    for tag in "abc":
        layers = [int(v) for v in z["layers_" + tag]]
        np.random.seed(int(z["seed_" + tag]))
        W = [2 * np.random.random((layers[i + 1], layers[i])) - 1 for i in range(len(layers) - 1)]
        B = [2 * np.random.random((layers[i + 1], 1)) - 1 for i in range(len(layers) - 1)]
        y, am = feed(layers, W, B, z["x_" + tag])
        np.testing.assert_allclose(y, z["y_" + tag], rtol=1e-12, atol=0)   # fp64, summation order differs
        assert np.array_equal(am, z["argmax_" + tag])
    rs = np.random.RandomState(4)
    layers = [300, 64, 9, 4]
    W = [rs.uniform(-1, 1, (layers[i + 1], layers[i])) for i in range(3)]
    B = [rs.uniform(-1, 1, (layers[i + 1], 1)) for i in range(3)]
    X = rs.uniform(-1, 1, (33, 300))
    y, _ = feed(layers, W, B, X)
    want = np.stack([pyoracle.nn_feed(layers, W, B, x) for x in X])
    np.testing.assert_allclose(y, want, rtol=1e-12, atol=0)
    b.close()


def test_scratch_nn_on_live_observation():
    """[Observation.size=320008, 9, 4] topology (battleground.py:55-57) fed with
    the live observation vector of every (arena, ship): sparse-gather first
    layer == dense oracle feed on the materialised toVector()."""
    from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
    from oracle import pyoracle
    N, M, seed = 6, 4, 9
    b = ArenaBatch(N, M)
    b.spawn_random(seed)
    for t in range(40):
        b.bot_actions(["turret"] * M, seed, tick=t)
        b.step(actions_ptr=b._actions.ptr)
    layers = np.array([320008, 9, 4], np.int32)
    rs = np.random.RandomState(0)
    W = [2 * rs.random_sample((9, 320008)) - 1, 2 * rs.random_sample((4, 9)) - 1]
    B = [2 * rs.random_sample((9, 1)) - 1, 2 * rs.random_sample((4, 1)) - 1]
    w = np.concatenate([a.ravel() for a in W]); bb = np.concatenate([a.ravel() for a in B])
    dw, db = DeviceBuffer(w.nbytes).upload(w), DeviceBuffer(bb.nbytes).upload(bb)
    dy, da = DeviceBuffer(8 * N * M * 4), DeviceBuffer(4 * N * M)
    nat.check(nat.lib().ofx_scratch_feed_obs(b.handle, layers.ctypes.data_as(C.c_void_p), 3, dw.ptr, db.ptr,
                                             dy.ptr, da.ptr))
    b.sync()
    y = dy.download(np.float64, (N, M, 4)); am = da.download(np.int32, (N, M))
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_F64)
    for g in range(N):
        for i in range(M):
            vec = np.concatenate([head[g, i], sm[g].ravel(), lm[g].ravel()])   # observation.py:119-125
            want = pyoracle.nn_feed(layers, W, B, vec)
            np.testing.assert_allclose(y[g, i], want, rtol=1e-11, atol=0)
            assert am[g, i] == int(np.argmax(want))
    b.close()
