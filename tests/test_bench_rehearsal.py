"""The launch line the driver uses for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`)
rehearsed with two ranks on the ONE card of the test box (OFX_DIST_BACKEND=gloo: RCCL refuses two ranks per device):
rendezvous on 127.0.0.1, arena sharding by rank, the ShardedRollout loop with an episode end (restart + score
all-reduce) inside the timed region, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, nproc=None, env=None):
    e = dict(os.environ, **(env or {}))
    if nproc:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py"] + args
    else:
        cmd = [sys.executable, "bench.py"] + args
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@gpu
def test_two_rank_launch_line():
    d = _bench(["--gpus", "2", "--arenas", "64", "--steps", "24", "--warmup", "4", "--no-cpu-baseline"], nproc=2,
               env={"OFX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["arenas_per_gpu"] == 64
    assert d["config"]["episodes_in_timed_region"] >= 1          # restart + all-reduce were timed
    assert d["config"]["last_episode_arenas"] == 2 * 64          # the count went through the all-reduce
    assert d["roofline"]["kernel"] == "k_head_stream" and d["roofline"]["avg_kernel_ms"] > 0


@gpu
def test_single_rank_line_has_the_contract_fields():
    d = _bench(["--arenas", "128", "--steps", "12", "--warmup", "3", "--no-extra"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["cpu_baseline"]["all_cores"]["cores"] >= 1 and d["cpu_baseline"]["cpu_model"]
    assert d["config"]["episodes_in_timed_region"] >= 1


@gpu
def test_two_rank_strong_scaling_line():
    """--scaling strong: a FIXED total (BASELINE configs[4] uses 32768; 256 here) split over the ranks by global arena id."""
    d = _bench(["--gpus", "2", "--scaling", "strong", "--total-arenas", "256", "--steps", "24", "--warmup", "4",
                "--no-cpu-baseline"], nproc=2, env={"OFX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["arenas_per_gpu"] == 128
    assert d["config"]["episodes_in_timed_region"] >= 1
    assert d["config"]["last_episode_arenas"] == 256             # both shards' counts went through the all-reduce


@gpu
def test_gpus_flag_alone_launches_the_ranks():
    """`python bench.py --gpus 2` with NO launcher in the test's own command: the script starts its two ranks itself
    (before it touches the GPU) and relays rank 0's one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OFX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--arenas", "64", "--steps", "24", "--warmup", "4",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["arenas_per_gpu"] == 64
    assert d["config"]["last_episode_arenas"] == 128


def _refusal(args, env_add):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_add)
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=120)


def test_world_size_mismatch_refuses_without_a_line():
    """A rank whose WORLD_SIZE is not --gpus must not print a line that could be read as an N-GPU number (CPU: the
    check runs before torch or the library is imported)."""
    p = _refusal(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = _refusal(["--gpus", "1", "--steps", "2", "--warmup", "1"], {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_launcher_parent_starts_n_ranks_and_returns_their_code(tmp_path, monkeypatch):
    """The parent of `python bench.py --gpus N`: builds the torch.distributed.run line for N ranks with the caller's
    arguments, imports no torch, returns the workers' exit code."""
    import importlib
    sys.modules.pop("bench", None)
    had_torch = "torch" in sys.modules
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import subprocess as sp
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"])
    assert rc == 7
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and int(c[c.index("--master-port") + 1]) > 0
    assert c[-4:] == ["--gpus", "4", "--steps", "3"] and c[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert had_torch or "torch" not in sys.modules
