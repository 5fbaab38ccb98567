"""Pins the CPU oracle (oracle/ofx_oracle.c) against fixtures captured from the
live reference (oracle/gen_golden.py).  Everything here is bit-exact: ship
coordinates are ints, laser coordinates are accumulated doubles compared with
==, maps are binary."""
import numpy as np
import pytest

from oracle import pyoracle
from tests.trace_util import load_trace, step_traces, unpack_map, GOLDEN


@pytest.mark.parametrize("name", step_traces())
def test_step_trace(name):
    z = load_trace(name)
    M = z["init_state"].shape[0]
    ticks, episodes = int(z["ticks"]), int(z["episodes"])
    a = pyoracle.Arena(n_ships=M)
    a.spawn(z["spawn_draws"])
    for i, (x, y, px, py) in enumerate(z["init_state"]):
        a.set_ship(i, x, y, px, py)
    sm, _ = a.rasterise()
    assert np.array_equal(sm, unpack_map(z["init_ship_map"]))
    map_idx = {int(t): k for k, t in enumerate(z["map_ticks"])}
    t = 0
    for ep in range(episodes):
        for _ in range(ticks):
            head, done = a.obs_head()
            assert np.array_equal(head, z["obs8"][t]), (name, t, "obs head")
            assert np.array_equal(done, z["obs_done"][t]), (name, t, "done")
            act = z["actions"][t]
            a.step(act)
            s = a.ships()
            assert np.array_equal(s["xy"], z["ship_xy"][t]), (name, t, "ship xy")
            # a ship's pointing only changes when it acts; the trace stores it for all
            assert np.array_equal(s["pt"], z["ship_pt"][t]), (name, t, "pointing")
            assert np.array_equal(s["alive"], z["ship_alive"][t]), (name, t, "alive")
            assert np.array_equal(s["reward"], z["reward"][t]), (name, t, "reward")
            assert np.array_equal(s["score"], z["score"][t]), (name, t, "score")
            l = a.lasers()
            n = int(z["n_lasers"][t])
            assert len(l["x"]) == n, (name, t, "n_lasers", len(l["x"]), n)
            assert np.array_equal(l["x"], z["laser_x"][t, :n]), (name, t, "laser x")
            assert np.array_equal(l["y"], z["laser_y"][t, :n]), (name, t, "laser y")
            assert np.array_equal(l["owner"], z["laser_owner"][t, :n]), (name, t, "owner")
            assert np.array_equal(l["destroyed"], z["laser_destroyed"][t, :n]), (name, t, "destroyed")
            if t in map_idx:
                sm, lm = a.rasterise()
                k = map_idx[t]
                assert np.array_equal(sm, unpack_map(z["ship_maps"][k])), (name, t, "ship map")
                assert np.array_equal(lm, unpack_map(z["laser_maps"][k])), (name, t, "laser map")
            t += 1
        a.restart(z["reset_draws"][ep])
        s = a.ships()
        assert np.array_equal(np.concatenate([s["xy"], s["pt"]], axis=1), z["reset_state"][ep]), (name, ep, "reset")
        assert np.array_equal(s["last_score"], z["ep_scores"][ep]), (name, ep, "scores")
        assert np.all(s["alive"] == 1) and np.all(s["score"] == 0)
        assert len(a.lasers()["x"]) == 0


def test_raster_cases():
    zf = np.load(GOLDEN + "/raster_cases.npz")
    z = {k: zf[k] for k in zf.files}
    for (x, y, r), bits in zip(z["cases"], z["maps"]):
        cx, cy = (float(int(x)), float(int(y))) if r == 8.0 else (x, y)
        got = pyoracle.disk(cy, cx, r)          # disk((y, x)): row = y
        assert np.array_equal(got, unpack_map(bits)), (x, y, r)


def test_geometry_cases():
    zf = np.load(GOLDEN + "/geometry.npz")
    z = {k: zf[k] for k in zf.files}   # NpzFile re-inflates on every access
    cfg = pyoracle.default_cfg()
    for k in range(len(z["inp"])):
        sx, sy, px, py, ex, ey = (int(v) for v in z["inp"][k])
        assert pyoracle.enemy_aimed(cfg, px, py, ex, ey) == bool(z["aimed"][k]), k
        assert pyoracle.enemy_on_trajectory(cfg, sx, sy, px, py, ex, ey) == bool(z["traj"][k]), k
        e = pyoracle.edge(sx, sy, 8, px, py, 2)
        want = None if z["edge"][k, 0] == 0 else (int(z["edge"][k, 1]), int(z["edge"][k, 2]))
        assert e == want, k
        assert pyoracle.thrust(cfg, sx, sy, px, py) == (int(z["thrust"][k, 0]), int(z["thrust"][k, 1])), k
        # fired = ship centre rule, ship.py:147-148
        d2 = (sx - px) ** 2 + (sy - py) ** 2
        assert (np.sqrt(float(d2)) <= 10.0) == bool(z["fired_centre"][k]), k


def test_angle_with_comment_values():
    """form.py:341-358 documents angle_with for the 8 compass points as comments:
    0, pi/4, pi/2, 3pi/4, pi, -3pi/4, -pi/2, -pi/4.  The trajectory cone shifts
    them by +pi; probe that shift through enemy_on_trajectory: a target in the
    pointing direction is always inside its own cone unless the cone wraps."""
    cfg = pyoracle.default_cfg()
    dirs = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]
    got = [pyoracle.enemy_on_trajectory(cfg, 200, 200, 200 + 50 * dx, 200 + 50 * dy, 200 + 90 * dx, 200 + 90 * dy)
           for dx, dy in dirs]
    # (-1, 0): atan2(0,-1)+pi = 2pi -> cone wraps past 2pi -> never a hit (ship.py:203-208)
    assert got == [True, True, True, True, False, True, True, True]


def test_scratch_nn():
    zf = np.load(GOLDEN + "/scratch_nn.npz")
    z = {k: zf[k] for k in zf.files}
    for tag in "abc":
        layers = [int(v) for v in z["layers_" + tag]]
        np.random.seed(int(z["seed_" + tag]))
        # neural_network.py:108-111: weights first (all layers), then biases
        W = [2 * np.random.random((layers[i + 1], layers[i])) - 1 for i in range(len(layers) - 1)]
        B = [2 * np.random.random((layers[i + 1], 1)) - 1 for i in range(len(layers) - 1)]
        for x, y, am in zip(z["x_" + tag], z["y_" + tag], z["argmax_" + tag]):
            got = pyoracle.nn_feed(layers, W, B, x)
            np.testing.assert_allclose(got, y, rtol=1e-12, atol=0)
            assert int(np.argmax(got)) == int(am)
