"""bench.py --gpus N must start by itself from a bare shell (VERDICT r1 missing-1): the parent spawns
``torch.distributed.run`` with N fresh ranks before anything touches the GPU.  Rehearsed here on CPU: ``--dry --backend gloo``
goes through the same launcher and rendezvous and prints the one JSON line with the number of ranks the group saw."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--backend", "gloo", "--steps", "1", "--warmup", "0"] + extra,
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["allreduce_sum"] == 3.0
    # the per-rank diagnostics of a multi-rank line: one entry per rank in rank order + min / median / max
    pr = out["per_rank"]
    assert pr["ms_per_step"] == [10.0, 11.0] and pr["exposed_collective_ms_per_step"] == [0.0, 0.5]
    assert pr["scored_rows_per_step"] == [1000.0, 2000.0] and pr["ms_per_step_min_median_max"] == [10.0, 10.5, 11.0]
    assert pr["compute_ms_per_step"] == [10.0, 10.5]
    usable = len(os.sched_getaffinity(0))
    assert out["host_threads_per_rank"] == max(1, usable // 2), "host threads default to this node's cores // ranks"


def test_bench_host_threads_flag():
    assert _run(["--gpus", "2", "--host-threads", "3"])["host_threads_per_rank"] == 3


def test_bench_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1"])
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1


def test_bench_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
