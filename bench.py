#!/usr/bin/env python
"""bench.py - arena-steps/sec of the hot path on N MI355X (one process per GPU).

A "step" is one lock-step of the per-GPU arena batch: on-device random-bot
actions (agents/agent.py:123-133 law, counter RNG) -> ofx_step (physics +
collision + reward) -> ofx_rasterise (u8 ship/laser maps) [-> policy forward,
once built], with Battleground.restart + the episodic score all-reduce (RCCL)
every 200 ticks.  Arenas shard by global id (weak scaling: 4096 per GPU); the
only collective is the [M+1] int64 score all-reduce at episode ends.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     - dominant kernel's algorithmic bytes / measured HIP-event time
  cpu_baseline - the CPU oracle timed on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x0F160001
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_policy_sample(pyoracle, cfg, w, n_arenas, ticks, n_pol, seed):
    """CPU-oracle leg of the full workload: per tick and arena, n_pol policy forwards (the oracle has no
    trunk sharing: one full forward per ship, like the reference), step, rasterise."""
    import numpy as np
    M = cfg.n_ships
    for g in range(n_arenas):
        a = pyoracle.Arena(cfg=cfg)
        a.spawn(pyoracle.reset_draws(cfg, seed, g, 0))
        for t in range(ticks):
            sm, lm = a.rasterise()
            head, _ = a.obs_head()
            act = a.bot_actions(np.ones(M, np.int32), seed, g, t)
            for i in range(n_pol):
                _, _, ia, ip = pyoracle.policy_forward(sm, lm, head[i].astype(np.float32), w, want_heat=False)
                act[i, 1], act[i, 2], act[i, 3], act[i, 4] = int(ia == 0), int(ia == 1), ip[0], ip[1]
            a.step(act)


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC summary (profiles/r01_full_pmc_hbm.txt: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command, tools/final_profile.sh; counters cannot be
    collected from inside a timed run).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.
    None when the file or the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "r01_full_pmc_hbm.txt")
    try:
        kb = {}
        for line in open(path):
            f = line.rstrip("\n").split("\t")
            if len(f) == 2 and f[0].startswith(kernel_prefix) and "=" in f[1]:
                name, val = f[1].split("=")
                kb[name] = float(val)
        if "FETCH_SIZE" in kb and "WRITE_SIZE" in kb:
            return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0
    except OSError:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--arenas", type=int, default=4096, help="arenas per GPU")
    ap.add_argument("--ships", type=int, default=8)
    ap.add_argument("--workload", default="step+obs+policy", choices=["step", "step+obs", "step+obs+policy"])
    ap.add_argument("--policy-ships", type=int, default=-1, help="ships per arena driven by the bi-head policy (-1 = all)")
    ap.add_argument("--policy-alive-only", action="store_true",
                    help="skip the forward of destroyed ships (QlearnIA.play returns None once done, "
                         "agents/qlearnIA_V2.py:372-377); NOT the headline configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch

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

    N, M = args.arenas, args.ships
    b = ArenaBatch(N, M, device=local_rank, arena_base=rank * N)
    beh = ["random"] * M
    ep_ticks = b.cfg.episode_ticks
    do_obs = args.workload != "step"
    do_policy = args.workload == "step+obs+policy"
    n_pol = M if args.policy_ships < 0 else min(M, args.policy_ships)
    if do_policy:
        # synthetic he_uniform / glorot_uniform weights of the pointer_model architecture (no checkpoint
        # ships with the reference); seed 0x0F160002
        from ofighters_amd.agents.policy_weights import synthetic
        w_host = synthetic(0x0F160002)
        w_dev = torch.from_numpy(w_host).cuda()
        mask_dev = None
        if n_pol < M:
            mk = np.zeros((N, M), np.uint8)
            mk[:, :n_pol] = 1
            mask_dev = torch.from_numpy(mk).cuda()
        if args.policy_alive_only:
            # the engine's own alive flags [N][M] uint8 serve as the ship mask: zero extra work
            class _AliveMask:
                def data_ptr(self_inner):
                    return b.device_ptr(nat.F_SHIP_ALIVE)
            mask_dev = _AliveMask()
    scores = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()  # the handle's stream is non-blocking w.r.t. torch's
    score_log = []

    def episode_end():
        # Agent.reset banks the episode score (agents/agent.py:61-63); the
        # per-slot sums over all arenas of all GPUs are the one cross-GPU number.
        b.restart_random(SEED)
        b.episode_scores_into(scores.data_ptr())
        b.sync()
        if dist is not None:
            dist.all_reduce(scores)
        score_log.append(scores.clone())
        torch.cuda.synchronize()

    tick = [0]
    EV = [0]

    def lockstep(timed):
        t = tick[0]
        if t > 0 and t % ep_ticks == 0:
            episode_end()
        b.bot_actions(beh, SEED, tick=t)
        if do_policy:
            # request_actions: the policy ships' actions overwrite the scripted ones
            if timed and EV[0] == 0:
                b.policy_profile(0)     # the library brackets its dominant kernel with event pairs from here on
            b.policy_forward(w_dev.data_ptr(), mask_dev.data_ptr() if mask_dev is not None else None)
            if timed:
                EV[0] += 2
            b.policy_actions(ship_mask_ptr=mask_dev.data_ptr() if mask_dev is not None else None)
        if timed and not do_obs:
            b.event_record(EV[0])
        b.step(actions_ptr=b._actions.ptr)
        if timed and not do_obs:
            b.event_record(EV[0] + 1)
            EV[0] += 2
        if do_obs:
            if timed and not do_policy:
                b.event_record(EV[0])
            b.rasterise(nat.MAP_U8)
            if timed and not do_policy:
                b.event_record(EV[0] + 1)
                EV[0] += 2
        tick[0] = t + 1

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    b.spawn_random(SEED)
    for _ in range(args.warmup):
        lockstep(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lockstep(True)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # dominant kernel: average launch duration from the HIP events recorded on
    # the handle's stream inside the timed region
    n_ev = EV[0] // 2
    k_ms = [b.event_elapsed(2 * i, 2 * i + 1) for i in range(min(n_ev, 32000))]
    k_avg_ms = float(np.mean(k_ms)) if k_ms else float("nan")
    overflow = b.overflow_count()
    roof_unit, roof_peak, roof_bound, roof_note = "GB/s", HBM_PEAK_GBS, "hbm", None
    if do_policy:
        # dense algorithmic FLOPs, no sparsity credit: trunk 53.28 MMAC once per arena +
        # 24.37 MMAC per policy ship
        # dominant kernel: k_head_tail = the last three up-convolutions of head-2 + arg-max, fused:
        # [x2 bilinear + conv 2->4 @100^2] + [x2 + conv 4->8 @200^2] + [x2 + conv 8->1 @400^2]
        # per policy ship: 0.72 + 11.52 + 11.52 = 23.76 MMAC, dense, no sparsity credit
        kernel = "k_head_tail"
        n_eff = n_pol
        if args.policy_alive_only:   # forwards actually run ~ mean number of playable ships (sampled at the end)
            n_eff = float(b.get(nat.F_SHIP_ALIVE).mean()) * M
        alg_flops = N * n_eff * 2.0 * 23.76e6
        whole_forward_flops = N * 2.0 * (53.28e6 + n_eff * 24.37e6)
        achieved = alg_flops / (k_avg_ms * 1e-3) / 1e12
        roof_unit, roof_peak, roof_bound = "TFLOP/s", 157.3, "mfma"
        roof_note = ("fp32 (exact f32-input MFMA + fp32 VALU, both 157.3 TFLOP/s peak); whole tick = %.0f GFLOP dense "
                     "algorithmic (trunk once per arena + per-ship heads) = %.1f TFLOP/s over ms_per_step"
                     % (whole_forward_flops / 1e9, whole_forward_flops / (dt / args.steps) / 1e12))
        alg_bytes = None
    elif do_obs:
        kernel = "k_raster<u8>"
        alg_bytes = N * 2 * b.W * b.H * 1          # two u8 maps written per arena (SURVEY 8d cfg 3)
    else:
        kernel = "k_step"
        alg_bytes = N * 3200                        # SURVEY 8d cfg 2: ~3.2 KB per arena-step
    if not do_policy:
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9

    out = {
        "metric": "arena-steps/sec (env.step+obs+policy fwd) at 4096 arenas, 1/2/4/8 MI355X",
        "value": world * N * args.steps / dt,
        "unit": "arena-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if do_policy else "f64",
        "data": "synthetic",
        "config": {
            "workload": ("%d arenas x %d ships per GPU, random-bot actions + step" % (N, M))
                        + (" + 2D obs rasterise (u8 maps)" if do_obs else "")
                        + ((" + bi-head policy forward for %d ship(s)/arena, trunk shared per arena (BASELINE configs[3])"
                            % n_pol) + (" (destroyed ships skipped)" if do_policy and args.policy_alive_only else "")
                           if do_policy else "; no policy forward (BASELINE configs[%d])" % (2 if do_obs else 1)),
            "arenas_per_gpu": N, "ships": M, "laser_cap": b.L, "episode_ticks": ep_ticks,
            "parallelism": "arena-sharded x%d, RCCL all-reduce of episodic scores only" % world,
            "laser_overflow": overflow,
        },
        "roofline": {
            "bound": roof_bound, "kernel": kernel, "achieved": achieved, "peak": roof_peak, "unit": roof_unit,
            "frac": achieved / roof_peak,
            "traffic": pmc_traffic({"k_head_tail": "k_head_tail", "k_raster<u8>": "void k_raster<0>"}.get(kernel, "\0"))
                       if (N == 4096 and M == 8 and (not do_policy or (n_pol == M and not args.policy_alive_only))) else None,
            "avg_kernel_ms": k_avg_ms,
            ("algorithmic_flops_per_launch" if do_policy else "algorithmic_bytes_per_launch"):
                (alg_flops if do_policy else alg_bytes),
        },
    }
    if roof_note:
        out["roofline"]["note"] = roof_note
    if out["roofline"]["traffic"] is not None:
        out["roofline"]["traffic_source"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate "
                                             "passes of this command: profiles/r01_full_pmc_hbm.txt")
    if score_log:
        out["config"]["last_episode_score_sum"] = int(score_log[-1][:M].sum().item())
        out["config"]["last_episode_arenas"] = int(score_log[-1][M].item())

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU oracle (a port, 1 thread) on a bounded sample of the same workload
        from oracle import pyoracle
        cfg = pyoracle.default_cfg(M)
        if do_policy:
            # one policy forward costs ~0.3 s on one core: sample = a few arena-steps of the full workload
            n_s, t_s = 2, 3
            c0 = time.perf_counter()
            cpu_policy_sample(pyoracle, cfg, w_host, n_s, t_s, n_pol, SEED)
            cdt = time.perf_counter() - c0
        else:
            n_s, t_s = (1024, 200) if do_obs else (4096, 200)
            c0 = time.perf_counter()
            pyoracle.run_random(cfg, n_s, t_s, SEED, int(do_obs), ep_ticks)
            cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {
            "value": n_s * t_s / cdt, "unit": "arena-steps/s", "cores": 1, "kind": "port",
            "sample": "%d arenas x %d ticks of the same workload (C oracle, single thread)" % (n_s, t_s),
        }
    if rank == 0:
        print(json.dumps(out))
    b.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
