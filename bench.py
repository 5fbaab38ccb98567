#!/usr/bin/env python
"""bench.py - arena-steps/sec of the hot path on N MI355X (one process per GPU).

A "step" is one lock-step of the per-GPU arena batch: on-device random-bot
actions (agents/agent.py:123-133 law, counter RNG) -> bi-head policy forward
for the policy ships -> ofx_step (physics + collision + reward) ->
ofx_rasterise (u8 ship/laser maps), with Battleground.restart + the episodic
score all-reduce (RCCL) every 200 ticks.  The loop is
ofighters_amd.rollout.ShardedRollout at EVERY world size - the class the
world_size-2 gloo test drives (tests/test_sharded_gloo.py).  Arenas shard by
global id: weak scaling (4096 per GPU, the default) or --scaling strong (32768
in total); the only collective is the [M+1] int64 score all-reduce at episode
ends.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline      - dominant kernel's algorithmic work / measured HIP-event time
  cpu_baseline  - the CPU oracle timed on a bounded sample (rank 0, N=1 only)
  extra_configs - at N=1: BASELINE configs[1] (step) and configs[2] (step+obs)
                  and the two reference-faithful policy workloads (one policy
                  ship per arena; destroyed ships skipped), each with its own
                  roofline - labelled secondary lines, never `value`
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x0F160001
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TF = 157.3    # fp32 vector = f32-input MFMA peak
METRIC = "arena-steps/sec (env.step+obs+policy fwd) at 4096 arenas, 1/2/4/8 MI355X"


def pmc_files():
    """The newest round's committed PMC summary + kernel stats (tools/final_profile.sh <tag> writes both; profiles/
    keeps them as r<NN>_full_pmc_hbm.txt / r<NN>_full_kernel_stats.csv)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_full_pmc_hbm.txt")):
        m = re.match(r"r(\d+)([a-z]?)_full_pmc_hbm\.txt$", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), m.group(2)) > best[0]):
            best = ((int(m.group(1)), m.group(2)), f)
    if best is None:
        return None, None
    return best[1], best[1].replace("_pmc_hbm.txt", "_kernel_stats.csv")


def pmc_traffic(kernel_prefix, live_ms=None):
    """(HBM bytes per launch of a kernel, source file) from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes of this same command, tools/final_profile.sh; counters cannot be collected from
    inside a timed run).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  (None, reason) when the
    file or the kernel is missing - e.g. after the kernel was renamed - or when the kernel's duration in that round's
    kernel stats is more than 15 % away from the one measured live (the counters describe another build): never a stale
    number."""
    import csv
    pmc, stats = pmc_files()
    if pmc is None:
        return None, "no profiles/r*_full_pmc_hbm.txt"
    try:
        kb = {}
        for line in open(pmc):
            f = line.rstrip("\n").split("\t")
            if len(f) == 2 and kernel_prefix in f[0] and "=" in f[1]:
                name, val = f[1].split("=")
                kb[name] = float(val)
        if not ("FETCH_SIZE" in kb and "WRITE_SIZE" in kb):
            return None, "kernel not in " + os.path.relpath(pmc, ROOT)
        if live_ms is not None:
            prof_ms = None
            for r in csv.DictReader(open(stats)):
                if kernel_prefix in r["Name"]:
                    prof_ms = float(r["AverageNs"]) / 1e6
                    break
            if prof_ms is None or abs(prof_ms - live_ms) > 0.15 * live_ms:
                return None, "stale: %s has %s ms for this kernel, live %.3f ms" % (os.path.relpath(stats, ROOT), prof_ms, live_ms)
        return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0, os.path.relpath(pmc, ROOT)
    except (OSError, KeyError, ValueError) as e:
        return None, "unreadable: %s" % e


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, M, n_pol, w_host, ep_ticks):
    """The C oracle (a port) on the box's host cores: one thread, then all cores - both as ONE C call
    (oracle/ofx_oracle.c orc_bench_run: POSIX threads over arenas, no Python in the loop).  Bounded samples of the same
    workload; the policy legs run one full forward per policy ship (no trunk sharing), like the reference."""
    from oracle import pyoracle
    cfg = pyoracle.default_cfg(M)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:  # a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>"): more threads than that only contend
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    if workload == "step+obs+policy":
        # ~0.27 s per arena-step with 8 forwards on one core: 64 arena-steps on one thread, 16 per thread on all cores
        n_one, t_one, n_all, t_all, obs, pol = 8, 8, 2 * cores, 8, 1, n_pol
    elif workload == "step+obs":
        n_one, t_one, n_all, t_all, obs, pol = 2048, 200, 512 * cores, 200, 1, 0
    else:
        n_one, t_one, n_all, t_all, obs, pol = 8192, 200, 2048 * cores, 200, 0, 0
    c0 = time.perf_counter()
    pyoracle.bench_run(cfg, w_host, n_one, t_one, SEED, obs, ep_ticks, pol, 1)
    d_one = time.perf_counter() - c0
    c0 = time.perf_counter()
    pyoracle.bench_run(cfg, w_host, n_all, t_all, SEED, obs, ep_ticks, pol, cores)
    d_all = time.perf_counter() - c0
    return {
        "value": n_one * t_one / d_one, "unit": "arena-steps/s", "cores": 1, "kind": "port",
        "sample": "%d arenas x %d ticks of the same workload (C oracle, single thread, %.1f s)" % (n_one, t_one, d_one),
        "all_cores": {"value": n_all * t_all / d_all, "cores": cores,
                      "sample": "%d arenas x %d ticks, %d POSIX threads over arenas in one C call, %.1f s"
                                % (n_all, t_all, cores, d_all)},
        "cpu_model": cpu_model(),
    }


def bf16_accuracy(w_host, n_arenas=8, m=8, ticks=12):
    """Error of the bf16-operand forward against a float64 evaluation of the declared graph (torch CPU ops,
    tests/policy_ref64.py - a checker, not a reference implementation) on n_arenas x m ships of a short rollout with the
    bench weights: max |heat - heat64| / max |heat64| and how often the pointer is the float64 map's arg-max; the fp32
    path's numbers on the same ships beside them."""
    import numpy as np
    import torch
    from ofighters_amd import ArenaBatch, _native as nat
    from tests import policy_ref64 as R
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    b = ArenaBatch(n_arenas, m)
    b.set_option(nat.OPT_TRUNK_FUSE, 1)      # the streaming trunk the 4096-arena workload runs (its conv2 / conv3 take the switch)
    b.spawn_random(SEED)
    b.rollout(["random"] * m, SEED, 0, ticks)
    out = {}
    for tag, flag in (("fp32", 0), ("bf16", 1), ("fp16", 2)):
        b.set_option(nat.OPT_POLICY_BF16, flag)
        out[tag] = b.policy_forward_host(w_host, want_heat=True)
    head, _ = b.observe_head()
    sm, lm = b.maps_host(nat.MAP_U8)
    rec = {t: {"heat_max_err_over_max": 0.0, "act_max_err_over_max": 0.0, "argmax_equal_float64": 0,
               "iaction_equal_float64": 0} for t in out}
    for g in range(n_arenas):
        a64, h64 = R.forward(sm[g], lm[g], head[g].astype(np.float32), w_host)
        for i in range(m):
            k = int(np.argmax(h64[i]))
            for t in out:
                ea = float(np.abs(out[t]["act"][g, i] - a64[i]).max() / max(1e-30, np.abs(a64[i]).max()))
                rec[t]["act_max_err_over_max"] = max(rec[t]["act_max_err_over_max"], ea)
                rec[t]["iaction_equal_float64"] += int(int(out[t]["iaction"][g, i]) == int(np.argmax(a64[i])))
                e = float(np.abs(out[t]["heat"][g, i] - h64[i]).max() / np.abs(h64[i]).max())
                rec[t]["heat_max_err_over_max"] = max(rec[t]["heat_max_err_over_max"], e)
                rec[t]["argmax_equal_float64"] += int(tuple(out[t]["ipointer"][g, i]) == (k % 400, k // 400))
    rec["ships"] = n_arenas * m
    for t in ("bf16", "fp16"):
        rec["argmax_%s_equal_fp32" % t] = int((out[t]["ipointer"] == out["fp32"]["ipointer"]).all(axis=2).sum())
    b.close()
    return rec


def train_tick(N, M, device, base, w_host, fence):
    """Secondary line (never `value`): the learning agent as a whole on the reference's own line-up shape (ONE policy
    ship per arena, lib/ofighters.py:53) - TrainingRollout = forward -> epsilon-greedy -> remember -> step -> rasterise
    with DeviceTrainer.replay on the reference's schedule (every 50 total steps and on lock-steps where a learning agent
    first sees its death, agents/qlearnIA_V2.py:376-378,414-415), the fit on `fit_batch` rows of the sampled minibatch.
    Reports ms per lock-step with the replays amortised in, the replays per lock-step and the ms of one replay."""
    from ofighters_amd import ArenaBatch
    from ofighters_amd.lib.epsilon import Epsilon_decay
    from ofighters_amd.rollout import TrainingRollout
    from ofighters_amd.trainer import DeviceTrainer
    b = ArenaBatch(N, M, device=device, arena_base=base)
    eps = Epsilon_decay()
    eps.set(0.1)
    tr = DeviceTrainer(b, w_host, epsilon=eps, batch_size=8, memory_size=64, frames=96, fit_batch=256)
    roll = TrainingRollout(b, tr, ["random"] * M, SEED, policy_ships=(0,), episode_ticks=b.cfg.episode_ticks)
    warm, steps = 30, 120
    roll.run(warm)
    fence()
    n0 = len(roll.losses)
    t0 = time.perf_counter()
    roll.run(steps)
    fence()
    dt = time.perf_counter() - t0
    n_rep = len(roll.losses) - n0
    t1 = time.perf_counter()
    for _ in range(3):
        tr.replay()
    fence()
    rep_ms = (time.perf_counter() - t1) / 3 * 1e3
    # the fit at the batch the rollout can feed it: 4096 rows of the sampled minibatch in ONE optimisation step
    big_ms = None
    try:
        tr.fit_batch = 4096
        tr.replay()
        fence()
        t2 = time.perf_counter()
        for _ in range(2):
            tr.replay()
        fence()
        big_ms = (time.perf_counter() - t2) / 2 * 1e3
    except Exception as e:
        big_ms = "failed: %s" % e
    rec = {"workload": "%d arenas x %d ships per GPU, TRAINING tick: one policy ship per arena (forward + epsilon-greedy + "
                       "transition capture) + step + obs + DeviceTrainer.replay on the reference's schedule "
                       "(every 50 steps or on lock-steps with a first-seen death: at most one per lock-step), fit_batch 256" % (N, M),
           "value": N * steps / dt, "unit": "arena-steps/s", "steps": steps, "warmup": warm,
           "ms_per_step": dt / steps * 1e3, "replays_in_timed_region": n_rep, "ms_per_replay": rep_ms,
           "last_losses": [float(x) for x in roll.losses[-3:]],
           "ms_per_replay_fit_batch_4096": big_ms,
           "note": "sample + gather + two target forwards + one fit step per replay; the fit is the lean form "
                   "(ofx_fit.hip: only the pre-activations kept, fused fp32 VALU tiles), not the hot path"}
    b.close()
    return rec


def scratch_feed_line(N, M, device, base, fence):
    """Secondary line (never `value`): P2, the scratch-NN forward (Neural_network.feed, agents/neural_network.py:396-420)
    with the topology the reference intends for it, [Observation.size = 320008, 9, Action.size = 4]
    (lib/battleground.py:55-57), fed with the LIVE observation of every (arena, ship): ofx_scratch_feed_obs = 1-bit
    rasterise + sparse gather-sum of W1 columns at the set cells (once per arena) + the 8-scalar head per ship + the
    dense 9 -> 4 layer + arg-max, float64.  Reports ms per call, the bytes it gathers from W1 and the fraction of the HBM
    roofline those algorithmic bytes (bit maps + gathered columns + outputs) would stand for."""
    import ctypes as C
    import numpy as np
    from ofighters_amd import ArenaBatch, DeviceBuffer, _native as nat
    b = ArenaBatch(N, M, device=device, arena_base=base)
    b.spawn_random(SEED)
    b.rollout(["random"] * M, SEED, 0, 100)                      # mid-episode state: ~25 live lasers per arena
    layers = np.array([8 + 2 * b.W * b.H, 9, 4], np.int32)
    rs = np.random.RandomState(0)
    w = np.concatenate([(2 * rs.random_sample(9 * int(layers[0])) - 1), (2 * rs.random_sample(36) - 1)])   # U[-1, 1), :108-111
    bb = 2 * rs.random_sample(13) - 1
    dw, db = DeviceBuffer(w.nbytes).upload(w), DeviceBuffer(bb.nbytes).upload(bb)
    dy, da = DeviceBuffer(8 * N * M * 4), DeviceBuffer(4 * N * M)
    lp = layers.ctypes.data_as(C.c_void_p)

    def call():
        nat.check(nat.lib().ofx_scratch_feed_obs(b.handle, lp, 3, dw.ptr, db.ptr, dy.ptr, da.ptr))
    for _ in range(5):
        call()
    fence()
    reps = 50
    b.timer_start()
    for _ in range(reps):
        call()
    ms = b.timer_stop() / reps
    sm, lm = b.maps_host(nat.MAP_BITS)
    set_cells = int(np.unpackbits(sm).sum()) + int(np.unpackbits(lm).sum())
    gathered = set_cells * 9 * 8
    alg = N * 2 * (b.W * b.H // 8) * 2 + gathered + N * M * (4 * 8 + 4)     # bit maps written + read, W1 columns, y + arg-max
    rec = {"workload": "%d arenas x %d ships per GPU: scratch-NN forward [320008, 9, 4] (float64) on the live observation "
                       "of every ship: 1-bit rasterise + sparse gather of W1 columns + head + 9 -> 4 + arg-max" % (N, M),
           "value": N / (ms * 1e-3), "unit": "arena-steps/s (forwards of all %d ships)" % M, "ms_per_call": ms,
           "set_cells_per_arena": set_cells / N, "bytes_gathered_from_W1": gathered,
           "roofline": {"bound": "hbm", "kernel": "k_obs_tail (+ k_raster<bits>, k_obs_first, k_mlp_layer)",
                        "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": alg,
                        "note": "W1 is 23 MB: the gathered columns come out of L2 / Infinity Cache, not HBM - the HBM "
                                "fraction is what the line's bytes would be worth at 8 TB/s, the kernel is latency-bound"}}
    b.close()
    return rec


class Workload:
    """One timed configuration on an ArenaBatch: ShardedRollout + the policy hook + HIP-event bracketing of the
    dominant kernel."""

    def __init__(self, b, workload, n_pol, alive_only, w_dev, dist, torch, nat, ep_ticks, start_tick=0):
        import numpy as np
        from ofighters_amd.rollout import ShardedRollout
        self.b, self.workload, self.n_pol, self.alive_only = b, workload, n_pol, alive_only
        self.do_obs = workload != "step"
        self.do_policy = workload == "step+obs+policy"
        self.nat, self.np = nat, np
        self.ev = 0
        self.timed = False
        N, M = b.N, b.M
        self.mask_ptr = None
        self._keep = None
        if self.do_policy:
            if alive_only:
                self.mask_ptr = b.device_ptr(nat.F_SHIP_ALIVE)   # the engine's alive flags [N][M] uint8 are the mask
            elif n_pol < M:
                mk = np.zeros((N, M), np.uint8)
                mk[:, :n_pol] = 1
                self._keep = torch.from_numpy(mk).cuda()
                self.mask_ptr = self._keep.data_ptr()
            b.policy_pin_weights(w_dev.data_ptr())               # prepared once: Keras keeps its compiled model too
        self.w_ptr = w_dev.data_ptr() if w_dev is not None else None
        self.stage = "policy" if self.do_policy else ("obs" if self.do_obs else "step")
        self.roll = ShardedRollout(b, ["random"] * M, SEED, episode_ticks=ep_ticks, dist=dist, observe=self.do_obs,
                                   policy=self._policy if self.do_policy else None,
                                   to_tensor=lambda a: torch.from_numpy(a.copy()).cuda(), start_tick=start_tick,
                                   probe=self._probe, use_rollout=not self.do_policy)
        self.chunks = []   # lock-steps per ofx_rollout call of the timed region (headless workloads)

    def _policy(self, e):
        # request_actions: the policy ships' actions overwrite the scripted ones
        if self.timed and self.ev == 0:
            e.policy_profile(0)          # the library brackets its dominant kernel with event pairs from here on
        e.policy_forward(self.w_ptr, self.mask_ptr)
        if self.timed:
            self.ev += 2
        e.policy_actions(ship_mask_ptr=self.mask_ptr)

    def _probe(self, stage, begin, n=1):
        # headless workloads (no policy): K lock-steps per ofx_rollout call; the library itself brackets the call's
        # dominant kernel with one event pair (the K-tick k_step launch / the last lock-step's k_raster)
        if self.timed and stage == self.stage and begin and self.ev < 60000:
            if self.ev == 0:
                self.b.policy_profile(0)
            self.chunks.append(n)
            self.ev += 2

    def run(self, warmup, steps, fence, torch, dist):
        self.roll.run(warmup)
        fence()
        self.episodes_before = len(self.roll.score_log)
        self.timed = True
        t0 = time.perf_counter()
        self.roll.run(steps)
        fence()
        dt = time.perf_counter() - t0
        self.timed = False
        self.b.policy_profile(-1)
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        n_ev = min(self.ev // 2, 30000)
        k_ms = [self.b.event_elapsed(2 * i, 2 * i + 1) for i in range(n_ev)]
        if not k_ms:
            return dt, float("nan")
        if self.stage == "step":     # one launch = chunks[i] lock-steps: report the time per lock-step
            return dt, float(self.np.sum(k_ms) / max(1, sum(self.chunks[:n_ev])))
        return dt, float(self.np.mean(k_ms))

    def roofline(self, k_avg_ms, dt, steps, full_config):
        b, np, nat = self.b, self.np, self.nat
        N, M = b.N, b.M
        if self.do_policy:
            # dominant kernel: k_head_stream = the last three up-convolutions of head-2 + arg-max, fused:
            # [x2 bilinear + conv 2->4 @100^2] + [x2 + conv 4->8 @200^2] + [x2 + conv 8->1 @400^2]
            # per policy ship: 0.72 + 11.52 + 11.52 = 23.76 MMAC, dense, no sparsity credit
            n_eff = self.n_pol
            if self.alive_only:  # forwards actually run ~ mean number of playable ships (sampled at the end)
                n_eff = float(b.get(nat.F_SHIP_ALIVE).mean()) * M
            alg = N * n_eff * 2.0 * 23.76e6
            whole = N * 2.0 * (53.28e6 + n_eff * 24.37e6)   # trunk once per arena + per-ship heads
            ach = alg / (k_avg_ms * 1e-3) / 1e12
            r = {"bound": "mfma", "kernel": "k_head_stream", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                 "frac": ach / FP32_PEAK_TF,
                 "traffic": None,
                 "avg_kernel_ms": k_avg_ms, "algorithmic_flops_per_launch": alg,
                 "note": "fp32 (exact f32-input MFMA + fp32 VALU, both 157.3 TFLOP/s peak); whole tick = %.0f GFLOP dense "
                         "algorithmic (trunk once per arena + per-ship heads) = %.1f TFLOP/s over ms_per_step = %.3f of peak"
                         % (whole / 1e9, whole / (dt / steps) / 1e12, whole / (dt / steps) / 1e12 / FP32_PEAK_TF)}
        else:
            if self.do_obs:
                kernel, alg = "k_raster<u8>", N * 2 * b.W * b.H * 1   # two u8 maps written per arena (SURVEY 8d cfg 3)
            else:
                kernel, alg = "k_step<bots>", N * 3200                 # SURVEY 8d cfg 2: ~3.2 KB per arena-step
            ach = alg / (k_avg_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": ach / HBM_PEAK_GBS,
                 "traffic": None,
                 "avg_kernel_ms": k_avg_ms, "algorithmic_bytes_per_launch": alg}
            if not self.do_obs:
                r["note"] = ("ofx_rollout: all lock-steps up to the next episode end run inside ONE k_step launch (bots' law "
                             "in the kernel); avg_kernel_ms / algorithmic bytes are per lock-step = launch time / its "
                             "lock-steps (launches of %s lock-steps)" % sorted(set(self.chunks)))
        if full_config and (self.do_policy or self.do_obs):
            r["traffic"], src = pmc_traffic("k_head_stream" if self.do_policy else "k_raster<0>", k_avg_ms)
            r["traffic_source"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate "
                                   "passes of this command: " + src) if r["traffic"] is not None else "none (%s)" % src
        return r

    def describe(self):
        N, M = self.b.N, self.b.M
        s = "%d arenas x %d ships per GPU, random-bot actions + step" % (N, M)
        if self.do_obs:
            s += " + 2D obs rasterise (u8 maps)"
        if self.do_policy:
            s += (" + bi-head policy forward for %d ship(s)/arena, trunk shared per arena (BASELINE configs[3])" % self.n_pol)
            if self.alive_only:
                s += " (destroyed ships skipped)"
        else:
            s += "; no policy forward (BASELINE configs[%d])" % (2 if self.do_obs else 1)
        return s


def launch_ranks(n, argv):
    """Start n ranks of this script on this node (one process per GPU, rendezvous on 127.0.0.1, a free port) and wait
    for them.  Runs in a parent that has not imported torch: nothing here initialises the GPU.  Returns the exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--arenas", type=int, default=4096, help="arenas per GPU (weak scaling)")
    ap.add_argument("--ships", type=int, default=8)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: --total-arenas in all, split over the ranks (SURVEY 8d: fixed N = 32768)")
    ap.add_argument("--total-arenas", type=int, default=32768)
    ap.add_argument("--workload", default="step+obs+policy", choices=["step", "step+obs", "step+obs+policy"])
    ap.add_argument("--policy-ships", type=int, default=-1, help="ships per arena driven by the bi-head policy (-1 = all)")
    ap.add_argument("--policy-alive-only", action="store_true",
                    help="skip the forward of destroyed ships (QlearnIA.play returns None once done, "
                         "agents/qlearnIA_V2.py:372-377); NOT the headline configuration")
    ap.add_argument("--trunk-form", type=int, default=0, choices=[0, 1, 2],
                    help="OFX_OPT_TRUNK_FUSE for A/Bs of the trunk kernels (0 = the library's choice; results are "
                         "bit-identical in every form)")
    ap.add_argument("--policy-bf16", type=int, nargs="?", const=1, default=0, choices=[0, 1, 2],
                    help="OFX_OPT_POLICY_BF16 for the main line: 1 bf16, 2 fp16 operands (opt-in reduced precision: NOT the "
                         "headline configuration)")
    ap.add_argument("--trunk-sparse", action="store_true",
                    help="OFX_OPT_TRUNK_SPARSE for the main line (exact, bit-identical, opt-in: NOT the headline "
                         "configuration - for profiling the secondary lines)")
    ap.add_argument("--no-bf16-accuracy", dest="bf16_accuracy", action="store_false",
                    help="skip the float64 evaluation behind the bf16 lines' accuracy numbers (~20 s of CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations at N=1")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: THIS process becomes the launcher - it has imported neither
        # torch nor the library and never touches the GPU - and starts the N ranks as child processes through
        # torch.distributed.run (the line the driver itself uses for N > 1); rank 0's JSON line goes to the inherited
        # stdout, the exit code is the workers'.  No exec: the ranks are children, the parent waits.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line that says n_gpus = world while the caller asked for --gpus N would be read as an N-GPU number
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d (RANK %s): refusing to print a line for another "
                         "world size; launch %d ranks (python bench.py --gpus %d, or torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d)\n"
                         % (args.gpus, world, os.environ.get("RANK", "unset"), args.gpus, args.gpus, args.gpus, args.gpus))
        sys.exit(2)
    import numpy as np
    import torch
    n_dev = torch.cuda.device_count()
    if world > 1 and os.environ.get("OFX_DIST_BACKEND", "nccl") == "nccl" and n_dev < world:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s): RCCL needs one device per rank\n" % (world, n_dev))
        sys.exit(2)

    dist = None
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run: one process per GPU, RCCL ("nccl") over xGMI
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # OFX_DIST_BACKEND=gloo: rehearsal of the launch path with several ranks on ONE card (RCCL refuses two ranks
        # on the same device); the measured runs use RCCL
        backend = os.environ.get("OFX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ofighters_amd import ArenaBatch, _native as nat
    from ofighters_amd.rollout import shard_range

    M = args.ships
    if args.scaling == "strong":
        if args.total_arenas % world:
            raise SystemExit("--total-arenas %d is not a multiple of the world size %d" % (args.total_arenas, world))
        N = args.total_arenas // world
    else:
        N = args.arenas
    base, _ = shard_range(rank, world, N)
    n_pol = M if args.policy_ships < 0 else min(M, args.policy_ships)
    w_host = w_dev = None
    if args.workload == "step+obs+policy" or not args.no_extra:
        # synthetic he_uniform / glorot_uniform weights of the pointer_model architecture (no checkpoint
        # ships with the reference); seed 0x0F160002
        from ofighters_amd.agents.policy_weights import synthetic
        w_host = synthetic(0x0F160002)
        w_dev = torch.from_numpy(w_host).cuda()
    torch.cuda.synchronize()  # the handle's stream is non-blocking w.r.t. torch's

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(workload, n_pol_, alive_only, warmup, steps, full_config, bf16=False, sparse=False):
        b = ArenaBatch(N, M, device=local_rank, arena_base=base)
        if args.trunk_form:
            b.set_option(nat.OPT_TRUNK_FUSE, args.trunk_form)
        if bf16:
            b.set_option(nat.OPT_POLICY_BF16, int(bf16))
        if sparse:
            b.set_option(nat.OPT_TRUNK_SPARSE, 1)
        ep = b.cfg.episode_ticks
        # an episode end (restart + score all-reduce) falls into the middle of the timed region whatever --steps is
        start = (ep - warmup - max(1, steps // 2)) % ep
        wl = Workload(b, workload, n_pol_, alive_only, w_dev, dist, torch, nat, ep, start_tick=start)
        dt, k_ms = wl.run(warmup, steps, fence, torch, dist)
        rec = {"workload": wl.describe(), "value": world * N * steps / dt, "unit": "arena-steps/s", "steps": steps,
               "warmup": warmup, "ms_per_step": dt / steps * 1e3, "roofline": wl.roofline(k_ms, dt, steps, full_config),
               "laser_overflow": b.overflow_count(), "laser_cap": b.L, "episode_ticks": ep,
               "episodes_in_timed_region": len(wl.roll.score_log) - wl.episodes_before}
        if wl.roll.score_log:
            rec["last_episode_score_sum"] = int(wl.roll.score_log[-1][:M].sum())
            rec["last_episode_arenas"] = int(wl.roll.score_log[-1][M])
        if sparse:
            run, total, trun, ttotal = b.policy_trunk_stats()      # over warm-up + timed region
            frac = run / max(1, total)
            # dense algorithmic work of the tick minus the conv2 M-tiles that were not run (conv2 = 23.04 MMAC per arena);
            # the table passes that were skipped are LDS reads, not FLOPs
            n_eff = rec["roofline"]["algorithmic_flops_per_launch"] / (N * 2.0 * 23.76e6)
            whole = N * 2.0 * (53.28e6 + n_eff * 24.37e6)
            executed = whole - N * 2.0 * 23.04e6 * (1.0 - frac)
            rec["trunk_sparse"] = {"conv2_mtiles_executed_frac": frac, "conv1_table_passes_executed_frac": trun / max(1, ttotal),
                                   "executed_flops_per_tick": executed,
                                   "tick_tflops_on_executed_flops": executed / (dt / steps) / 1e12,
                                   "frac_of_fp32_peak_on_executed_flops": executed / (dt / steps) / 1e12 / FP32_PEAK_TF}
        b.close()
        return rec

    headline_full = N == 4096 and M == 8 and n_pol == M and not args.policy_alive_only and not args.policy_bf16 and not args.trunk_sparse
    head = measure(args.workload, n_pol, args.policy_alive_only, args.warmup, args.steps,
                   headline_full or args.workload != "step+obs+policy", bf16=args.policy_bf16, sparse=args.trunk_sparse)
    do_policy = args.workload == "step+obs+policy"
    out = {
        "metric": METRIC,
        "value": head["value"],
        "unit": "arena-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": (("bf16" if args.policy_bf16 == 1 else "fp16") + " operands, f32 accumulation (opt-in OFX_OPT_POLICY_BF16)" if args.policy_bf16 else "f32") if do_policy else "f64",
        "data": "synthetic",
        "config": {
            "workload": head["workload"],
            "arenas_per_gpu": N, "ships": M, "laser_cap": head["laser_cap"], "episode_ticks": head["episode_ticks"],
            "parallelism": "arena-sharded x%d, RCCL all-reduce of episodic scores only" % world,
            "laser_overflow": head["laser_overflow"],
            "episodes_in_timed_region": head["episodes_in_timed_region"],
        },
        "roofline": head["roofline"],
    }
    for k in ("last_episode_score_sum", "last_episode_arenas"):
        if k in head:
            out["config"][k] = head[k]

    if world == 1 and not args.no_extra and do_policy and headline_full:
        # secondary lines (never `value`): BASELINE configs[1] / [2], and the two reference-faithful policy workloads -
        # the stock line-up has ONE QlearnIA ship (lib/ofighters.py:53: bound 1.01 M arena-steps/s), and
        # QlearnIA.play returns None once the ship is destroyed (agents/qlearnIA_V2.py:372-377)
        extra = []
        for wl_name, np_, alive, wu, st in (("step", 0, False, 400, 20000), ("step+obs", 0, False, 200, 2000),
                                            ("step+obs+policy", 1, False, 50, 300), ("step+obs+policy", M, True, 50, 200)):
            r = measure(wl_name, np_, alive, wu, st, wl_name != "step+obs+policy")
            if wl_name == "step+obs+policy":
                flops = r["roofline"]["algorithmic_flops_per_launch"] / 23.76e6 * 24.37e6 + N * 2.0 * 53.28e6
                r["fp32_bound_arena_steps_per_s"] = N * FP32_PEAK_TF * 1e12 / flops
                r["frac_of_fp32_bound"] = r["value"] / r["fp32_bound_arena_steps_per_s"]
            extra.append(r)
        # OPT-IN, EXACT (OFX_OPT_TRUNK_SPARSE): the streaming trunk skips the constant windows of its ~1 %-dense input,
        # bit-identical results (tests/test_gpu_policy.py); the headline `value` above stays on the DENSE trunk this round
        for np_ in (M, 1):
            r = measure("step+obs+policy", np_, False, 40, 200, False, sparse=True)
            flops = r["roofline"]["algorithmic_flops_per_launch"] / 23.76e6 * 24.37e6 + N * 2.0 * 53.28e6
            r["fp32_bound_arena_steps_per_s"] = N * FP32_PEAK_TF * 1e12 / flops
            r["frac_of_fp32_bound"] = r["value"] / r["fp32_bound_arena_steps_per_s"]
            r["note"] = ("opt-in OFX_OPT_TRUNK_SPARSE = 1: exact (bit-identical outputs), fp32; a secondary line - the "
                         "headline runs the dense trunk")
            extra.append(r)
        # OPT-IN reduced precision (OFX_OPT_POLICY_BF16): never `value`, no parity and no roofline claim - the speed of the
        # bf16-operand head for 8 and 1 policy ships next to its measured error against the float64 graph
        acc = None
        if args.bf16_accuracy:
            try:
                acc = bf16_accuracy(w_host)
            except Exception as e:      # a secondary measurement must never take the headline line down with it
                acc = {"error": repr(e)}
        for lowp, tag in ((1, "bf16"), (2, "fp16")):
            for np_ in (M, 1):
                r = measure("step+obs+policy", np_, False, 30, 150, False, bf16=lowp)
                r["dtype"] = ("%s operands / fp32 accumulation in conv2-4 (streaming trunk) and upconv3-4 (97 %% of the per-ship "
                              "work); conv1 (exact table), dense layers, upconv1-2, frame lines fp32" % tag)
                r["north_star_target_arena_steps_per_s"] = 1.0e6
                r["frac_of_north_star_target"] = r["value"] / 1.0e6
                r["roofline"] = None
                r["note"] = "opt-in (OFX_OPT_POLICY_BF16 = %d), not the headline: earns no parity or roofline credit" % lowp
                if acc is not None:
                    r["accuracy_vs_float64"] = ({k: acc[k] for k in ("fp32", tag, "ships", "argmax_%s_equal_fp32" % tag)}
                                                if "error" not in acc else acc)
                extra.append(r)
        # both opt-ins together on the reference's own line-up: the line that passes north_star's 1 M arena-steps/s
        try:
            r = measure("step+obs+policy", 1, False, 30, 150, False, bf16=1, sparse=True)
            r["dtype"] = "bf16 operands / fp32 accumulation (OFX_OPT_POLICY_BF16 = 1) + the exact sparse trunk (OFX_OPT_TRUNK_SPARSE = 1)"
            r["north_star_target_arena_steps_per_s"] = 1.0e6
            r["frac_of_north_star_target"] = r["value"] / 1.0e6
            r["roofline"] = None
            r["note"] = "both opt-ins, one policy ship per arena: no parity or roofline credit (16-bit operands)"
            extra.append(r)
        except Exception as e:
            extra.append({"workload": "bf16 + sparse trunk", "error": repr(e)})
        try:
            extra.append(train_tick(N, M, local_rank, base, w_host, fence))
        except Exception as e:
            extra.append({"workload": "TRAINING tick", "error": repr(e)})
        try:
            extra.append(scratch_feed_line(N, M, local_rank, base, fence))
        except Exception as e:
            extra.append({"workload": "scratch-NN forward", "error": repr(e)})
        out["extra_configs"] = extra
        # the same secondary lines in compact form INSIDE `config` (a driver that keeps only the contract's keys still
        # sees BASELINE configs[1] / [2] and the reference-faithful line-ups): value + fraction of the line's own roofline
        def brief(r, frac_key=None):
            if "error" in r:
                return {"error": r["error"]}
            d = {"value": r["value"], "ms_per_step": r.get("ms_per_step", r.get("ms_per_call"))}
            if frac_key:
                d["frac"] = r.get(frac_key)
            elif r.get("roofline"):
                d["frac"], d["kernel"], d["kernel_ms"] = r["roofline"]["frac"], r["roofline"]["kernel"], r["roofline"].get("avg_kernel_ms")
            return d
        tt = extra[-2]
        out["config"]["secondary"] = {
            "step": brief(extra[0]), "step_obs": brief(extra[1]),
            "policy_1ship": brief(extra[2], "frac_of_fp32_bound"), "policy_alive_only": brief(extra[3], "frac_of_fp32_bound"),
            "sparse_trunk_8ships": brief(extra[4], "frac_of_fp32_bound"), "sparse_trunk_1ship": brief(extra[5], "frac_of_fp32_bound"),
            "sparse_trunk_conv2_mtiles_executed_frac": (extra[4].get("trunk_sparse") or {}).get("conv2_mtiles_executed_frac"),
            "bf16_8ships": brief(extra[6], "frac_of_north_star_target"), "bf16_1ship": brief(extra[7], "frac_of_north_star_target"),
            "bf16_sparse_trunk_1ship": brief(extra[10], "frac_of_north_star_target"),
            "train_tick_ms": tt.get("ms_per_step"), "fit4096_ms": tt.get("ms_per_replay_fit_batch_4096"),
            "scratch_feed_obs": brief(extra[-1]),
            "note": "secondary lines, never `value`; full records in extra_configs; sparse_trunk = opt-in exact OFX_OPT_TRUNK_SPARSE; "
                    "bf16 = opt-in OFX_OPT_POLICY_BF16 (no parity credit)",
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, M, n_pol, w_host, head["episode_ticks"])
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
