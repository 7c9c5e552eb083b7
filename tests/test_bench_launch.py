"""CPU: `python bench.py --gpus N` starts its N ranks itself (no launcher, WORLD_SIZE unset) and the combined-render leg runs at N > 1.
Rehearsed over gloo with CPU ops injected (`--dry-run-cpu`): the launch, the rendezvous on 127.0.0.1, the by-ray exchange, the gather and
the one JSON line of rank 0. The GPU form of the same leg is exercised on the box by bench.py itself (N = 1) and by the driver (N > 1)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks_prints_one_line_with_the_combined_leg():
    r = _run(2)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    leg = r["combined_render"]
    assert "error" not in leg and leg["objects"] == 2 and leg["value"] > 0
    assert leg["bytes_sent_per_view_per_gpu"] > 0 and leg["xgmi"]["peak_GBps_per_gpu"] == 153.0
    single = _run(1)
    assert single["n_gpus"] == 1 and single["combined_render"]["bytes_sent_per_view_per_gpu"] == 0


def test_self_launch_at_the_real_world_size_of_configs4():
    """`bench.py --gpus 8 --dry-run-cpu`: the launch the driver's SCALE run makes at N = 8 — eight gloo ranks, one object each, the by-ray
    exchange, the gather, ONE line from rank 0."""
    r = _run(8)
    assert r["n_gpus"] == 8 and r["scaling"] == "weak"
    leg = r["combined_render"]
    assert "error" not in leg and leg["objects"] == 8 and leg["world_size"] == 8 and leg["value"] > 0
    assert leg["bytes_sent_per_view_per_gpu"] > 0


def test_a_rank_failing_before_the_first_collective_fails_an_eight_rank_launch():
    """Rank 5 of 8 dies before the process group exists: the launcher tears the other seven down, the launch returns non-zero and NO result line is
    printed (a line without a measured leg would read as a measurement)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FOC_BENCH_FAIL_RANK"] = "5"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-1000:]


def test_a_failing_rank_fails_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FOC_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0


def test_a_stuck_exchange_leaves_one_line_and_a_nonzero_exit():
    """A rank that never reaches the exchange leaves the others waiting inside the collective: every rank gives the leg up on its own timer,
    rank 0 prints ONE line — the headline, with the leg's `error` — and the launch FAILS (bench.py ExchangeGuard: exit code 3, not 0)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FOC_BENCH_STALL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--exchange-timeout", "6"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0, p.stdout[-1000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and "no progress" in r["combined_render"]["error"] and r["combined_render"]["world_size"] == 2
