"""`python bench.py --gpus N` must start its own N ranks (the driver's SCALE run calls it both ways): the rank plumbing
-- spawn before any GPU call, 127.0.0.1 rendezvous, barrier / max-over-ranks, exactly ONE JSON line from rank 0, exit
code handed on -- is exercised here on the CPU with gloo (`--selftest-launch`: no model, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus2_spawns_its_own_ranks_and_prints_one_json_line():
    r = _run(["--gpus", "2", "--selftest-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_dt"] >= 0.02       # rank 1 sleeps 20 ms: MAX over ranks, not rank 0's own time


def test_under_torchrun_env_it_is_one_rank_and_rejects_a_world_mismatch():
    r = _run(["--gpus", "2", "--selftest-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_needs_no_launcher():
    r = _run(["--selftest-launch"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_failing_rank_makes_the_launcher_exit_non_zero_and_print_no_result_line():
    """`python bench.py --gpus 2`: the parent hands the ranks' exit status on (the driver reads it) -- a rank that dies must
    not leave a zero exit code or a JSON line behind"""
    r = _run(["--gpus", "2", "--selftest-launch", "--selftest-fail-rank", "1"])
    assert r.returncode != 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
