"""Experiment (GPU box): the headline tick as ONE batch of 4096 arenas against TWO half batches of 2048 arenas on their
own handles / HIP streams, launched alternately without a host sync - do the memory-bound kernels of one half (raster,
trunk tables, frames, upconv1) hide behind the other half's MFMA-bound k_head_stream?
usage: python tools/pipeline_probe.py [ticks]"""
import os
import sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofighters_amd import ArenaBatch, DeviceBuffer
from ofighters_amd.agents.policy_weights import synthetic

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M, SEED = 8, 0x0F160001
w = synthetic()


def make(n, base):
    b = ArenaBatch(n, M, arena_base=base)
    wd = DeviceBuffer(w.nbytes).upload(w)
    b.policy_pin_weights(wd.ptr)
    b.spawn_random(SEED)
    b.rasterise()
    return b, wd


def tick(b, wd, t):
    b.bot_actions(["random"] * M, SEED, tick=t)
    b.policy_forward(wd.ptr, None)
    b.policy_actions()
    b.step()
    b.rasterise()


def run(parts):
    for t in range(10):
        for b, wd in parts: tick(b, wd, t)
    for b, _ in parts: b.sync()
    t0 = time.perf_counter()
    for t in range(10, 10 + T):
        for b, wd in parts: tick(b, wd, t)
    for b, _ in parts: b.sync()
    return (time.perf_counter() - t0) / T * 1e3


one = [make(4096, 0)]
print("one batch of 4096 arenas      : %.3f ms per lock-step" % run(one), flush=True)
one[0][0].close()
two = [make(2048, 0), make(2048, 2048)]
print("two half batches, two streams : %.3f ms per lock-step of both" % run(two), flush=True)
for b, _ in two: b.close()
four = [make(1024, 1024 * i) for i in range(4)]
print("four quarter batches          : %.3f ms per lock-step of all" % run(four), flush=True)
for b, _ in four: b.close()
