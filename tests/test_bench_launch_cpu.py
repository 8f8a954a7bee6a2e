"""`python bench.py --gpus N` must be able to start its own ranks (it is the command the driver runs): the parent
spawns N children with the launcher environment, never touches the GPU itself, relays rank 0's JSON line and
exits with the worst child status.  Run here with the GPU work left out (--dry-run-ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TSR_CTRL_PORT", "TSR_CTRL_SECRET")}
    return env


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run-ranks"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["uid_ok"] and out["bank_templates"] == 2
    assert sorted(x[:2] for x in out["ranks"]) == [[0, 0], [1, 1]]
    assert len({x[2] for x in out["ranks"]}) == 2 and os.getpid() not in {x[2] for x in out["ranks"]}
    assert out["max_over_ranks_s"] == 0.002


def test_bench_parent_reports_the_worst_child_status():
    # WORLD_SIZE/--gpus disagreement inside the children cannot happen through the launcher; an unknown flag makes
    # every child exit 2 -- the parent must pass that on and print no JSON line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_bench_parent_does_not_load_hip_or_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("import tsr_hip")]
    assert "raise SystemExit(launch_ranks(" in head          # the launch happens before the library import
    assert "import torch" not in src


def test_bench_parent_gives_up_when_a_rank_dies_and_rank0_never_returns():
    # rank 1 exits early, rank 0 blocks (as inside a collective) and never closes its stdout: the parent's watchdog
    # must run beside the relay, terminate rank 0 by PID after the grace period and exit non-zero
    import time
    env = dict(_clean_env(), TSR_DRY_RUN_SCENARIO="rank1_dies_rank0_hangs", TSR_LAUNCH_GRACE_S="2")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-ranks"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert time.time() - t0 < 60
    assert "a rank failed" in r.stderr


def test_bench_parent_deadline():
    import time
    env = dict(_clean_env(), TSR_DRY_RUN_SCENARIO="rank1_dies_rank0_hangs", TSR_LAUNCH_GRACE_S="500",
               TSR_LAUNCH_DEADLINE_S="3")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-ranks"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60 and "deadline" in r.stderr
