"""Record / replay + headless frame dump: a recorded game replays bit for bit from its data
file; CPU test on the oracle engine, GPU test on libofx."""
import os
import random
import zlib

import numpy as np
import pytest

from ofighters_amd.lib.battleground import Battleground
from ofighters_amd.lib.record import OfighterRecord, render, save_png


def _play_and_replay(engine_factory, tmp_path):
    random.seed(5)
    bg = Battleground(ships={"random": 3, "turret": 1, "runner": 1}, engine=engine_factory(5))
    rec = OfighterRecord(bg, engine_factory=engine_factory)
    trace = []
    for ep in range(2):
        for _ in range(40):
            bg.frame()
            rec.saveFrame(bg.actions)
            trace.append(([(s.body.x, s.body.y, s.is_playable(), s.agent.reward, s.agent.score) for s in bg.ships],
                          [(l.body.x, l.body.y) for l in bg.lasers], bg.absolute_state.ship_map.copy()))
        if ep == 0:
            bg.restart()
            rec.saveRestart(bg)
    name = rec.save(os.path.join(str(tmp_path), "game"))
    rep = OfighterRecord.load(name, engine_factory=engine_factory)
    assert len(rep.actions) == 80 and len(rep.restarts) == 1
    for ships, lasers, ship_map in trace:
        obs = rep.nextFrame()
        assert [(s.body.x, s.body.y, s.is_playable(), s.agent.reward, s.agent.score) for s in rep.game.ships] == ships
        assert [(l.body.x, l.body.y) for l in rep.game.lasers] == lasers
        assert np.array_equal(obs.ship_map, ship_map)
    img = render(rep.game)
    assert img.shape == (400, 400, 3) and img.dtype == np.uint8
    assert (img.sum(axis=2) > 0).sum() == ((obs.ship_map != 0) | (obs.laser_map != 0)).sum()
    png = save_png(os.path.join(str(tmp_path), "frame.png"), img)
    data = open(png, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and b"IDAT" in data
    # the IDAT payload inflates to the raw scanlines
    i = data.index(b"IDAT")
    n = int.from_bytes(data[i - 4:i], "big")
    assert len(zlib.decompress(data[i + 4:i + 4 + n])) == 400 * (1 + 400 * 3)


def test_record_replay_cpu(tmp_path):
    from tests.oracle_batch import OracleEngine
    _play_and_replay(lambda M: OracleEngine(M), tmp_path)


@pytest.mark.gpu
def test_record_replay_gpu(tmp_path):
    from ofighters_amd import ArenaBatch
    _play_and_replay(lambda M: ArenaBatch(1, M), tmp_path)
