"""`python bench.py --gpus N` without torchrun around it must start its own ranks (BASELINE.json configs[3] is driven that
way): the parent spawns `python -m torch.distributed.run ... bench.py --gpus N ...` as a child BEFORE any GPU call, relays
rank 0's single JSON line and returns the children's status.  --launch-check runs that path with the GPU work replaced
by a gloo all-reduce, so the launcher is covered on a CPU-only box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_launch_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5] == BENCH and cmd[-4:] == ["--gpus", "8", "--steps", "5"]


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_spawns_ranks_and_relays_one_json_line(n):
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": n, "rank_sum": n * (n + 1) / 2}


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0 and json.loads(r.stdout.strip()) == {"launch_check": True, "n_gpus": 1, "rank_sum": 1.0}


def test_child_failure_is_the_parents_exit_status():
    r = _run(["--gpus", "2", "--launch-check", "--workload", "cfg3"], env_extra={"HMV_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not r.stdout.strip()


def test_gpus_must_match_world_size_under_torchrun():
    r = _run(["--gpus", "4", "--launch-check"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
