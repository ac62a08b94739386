"""bench.py's own multi-rank launcher (`python3 bench.py --gpus N` without an external torch.distributed.run): argument
and environment plumbing, rank 0's JSON relay and the exit code, on the CPU with gloo ranks (--launch-check; the GPU
ranks use the same launcher with the RCCL backend)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(n, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-check"], env=env, capture_output=True,
                          text=True, timeout=300)


def test_launcher_starts_n_ranks_and_relays_rank0_json():
    r = _run(3)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # one JSON line: rank 0's
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "world": 3, "rank_sum": 3.0, "local_rank": 0, "master_addr": "127.0.0.1"}


def test_launcher_exits_nonzero_when_a_rank_fails():
    r = _run(2, {"FU_BENCH_FAIL_RANK": "1", "FU_BENCH_LAUNCH_TIMEOUT": "60"})
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_single_rank_needs_no_launcher_and_parent_never_needs_a_gpu():
    # --gpus 1 falls through to the benchmark itself, which refuses to run without the MI355X (no CPU fallback);
    # with --gpus 2 the PARENT must get as far as starting ranks without any GPU (this container has none)
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
