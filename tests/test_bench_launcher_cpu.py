"""bench.py is its own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N ranks itself.
The rendezvous half runs here on CPU with gloo through the SAME launcher function and argument parser; the GPU half is
covered by tests/test_round3_gpu.py::test_bench_self_launch_two_gloo_ranks_one_gpu."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_bare_gpus_2_self_launches_and_all_ranks_meet():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--selftest-launcher"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["self_launched"] is True
    assert rec["sum_of_ranks_plus_1"] == 3.0 and abs(rec["max_time"] - 0.002) < 1e-12     # all-reduce SUM and MAX really ran
    assert rec["all_reduce_backend"] == "gloo"


def test_torchrun_style_environment_is_respected():
    """WORLD_SIZE already set (torchrun, the driver's form): no self-launch, the process IS a rank."""
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--selftest-launcher"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_ranks_seen"] == 1 and rec["self_launched"] is False


def test_world_size_mismatch_is_loud():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_dead_rank_takes_the_job_down():
    """A rank that dies must not leave the others waiting in the rendezvous forever."""
    sys.path.insert(0, REPO)
    import bench
    script = os.path.join(REPO, "tests", "_dying_rank.py")
    with open(script, "w") as f:
        f.write("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(120)\n")
    try:
        rc = bench.launch_ranks(2, [], script=script, timeout=60)
    finally:
        os.remove(script)
    assert rc != 0
