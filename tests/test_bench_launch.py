"""`python bench.py --gpus N` started plain (no launcher, no WORLD_SIZE) starts its own N ranks before it touches a
GPU and prints rank 0's one JSON line (bench.py: launch_ranks).  The reference has no multi-device code
(src/genjax/core.py:1304-1317 is dead plumbing): nothing to compare with, only the contract of the driver's command."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p, lines


def test_plain_start_with_two_ranks_meets_and_prints_one_line():
    """No GPU needed: the ranks only meet (gloo, 127.0.0.1) and rank 0 reports the world size."""
    p, lines = _run(["--gpus", "2", "--rendezvous-only"], 240)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 1


def test_a_failing_rank_fails_the_plain_start():
    """The children's exit status is the parent's: a bad argument in the ranks is not swallowed."""
    p, lines = _run(["--gpus", "2", "--particles", "1000", "--rendezvous-only", "--no-such-flag"], 240)
    assert p.returncode != 0 and not lines


@pytest.mark.gpu
def test_plain_start_runs_the_sharded_bench_on_two_gloo_ranks():
    """The driver's command shape at N = 2 on the one-GPU box: two gloo ranks share the GPU and run the real
    sharded step (HIP kernels, collectives staged through the host); one JSON line, n_gpus = 2, a finite value."""
    p, lines = _run(["--gpus", "2", "--backend", "gloo", "--particles", str(1 << 16), "--steps", "16", "--warmup", "17",
                     "--no-cpu-baseline"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 16 and line["value"] > 0
    assert line["config"]["particles_per_gpu"] == 1 << 15 and line["scaling"] == "strong"
