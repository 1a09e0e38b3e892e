"""CPU: `python bench.py --gpus N` starts N ranks itself (VERDICT r1 / ADVICE r1: --gpus was parsed and never used).
The parent never touches a GPU; children rendezvous on 127.0.0.1 (gloo here, RCCL on the GPU node), rank 0's JSON line is
relayed, and a failing rank makes the parent exit non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest-spawn"], env=env,
                          capture_output=True, text=True, timeout=300)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_sum"] == 3.0  # both ranks took part in the all-reduce: 1 + 2


def test_three_ranks():
    r = _run(3)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["ranks_sum"] == 6.0


def test_failing_rank_fails_the_job():
    r = _run(2, {"VMC_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode != 0


def test_rank_that_dies_before_the_rendezvous_ends_the_job_promptly():
    """A rank that exits before init_process_group (e.g. its device does not exist) leaves the others blocked in the rendezvous:
    the parent must end them (exact child PIDs) and fail, not wait for the collective time-out."""
    import time
    t0 = time.time()
    r = _run(2, {"VMC_SELFTEST_DIE_EARLY_RANK": "1"})
    assert r.returncode != 0
    assert time.time() - t0 < 90


def test_single_rank_needs_no_spawn():
    r = _run(1)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
