"""bench.py is its own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N ranks itself.
The rendezvous half runs here on CPU with gloo through the SAME launcher function and argument parser; the GPU half is
covered by tests/test_training_kernels_system_gpu.py::test_bench_self_launch_two_gloo_ranks_one_gpu."""
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


def test_a_rank_failing_in_setup_makes_every_rank_skip_the_record():
    """the multi-rank records wrapped collective-bearing code in a per-rank `except`; a rank failing alone left the others
    blocked in the record's first collective.  bench.setup_agreed runs the collective-FREE setup, then all ranks all-reduce a success flag:
    rank 1's failure becomes SetupFailed on BOTH ranks (here: two gloo ranks on CPU), and both are still able to run the next collective."""
    sys.path.insert(0, REPO)
    import bench
    script = os.path.join(REPO, "tests", "_setup_agreed_ranks.py")
    with open(script, "w") as f:
        f.write(
            "import os, sys, json\n"
            "sys.path.insert(0, %r)\n"
            "import torch, torch.distributed as dist, bench\n"
            "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            "dist.init_process_group('gloo')\n"
            "dev = torch.device('cpu')\n"
            "def setup():\n"
            "    if rank == 1:\n"
            "        raise MemoryError('rank 1 ran out of memory building its batch')\n"
            "    return 'ok'\n"
            "try:\n"
            "    bench.setup_agreed(setup, dev, world); got = 'entered'\n"
            "except bench.SetupFailed as e:\n"
            "    got = 'skipped: ' + str(e)\n"
            "ok = bench.setup_agreed(lambda: 42, dev, world)          # the next record's setup: everyone fine again\n"
            "t = torch.ones(1); dist.all_reduce(t)\n"
            "print(json.dumps({'rank': rank, 'got': got, 'next': ok, 'sum': float(t.item())}), file=sys.stderr, flush=True)\n"
            "dist.destroy_process_group()\n" % REPO)
    import io, contextlib
    try:
        out = subprocess.run([sys.executable, "-c",
                              "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(2, [], script=%r, timeout=120))" % (REPO, script)],
                             env=_clean_env(), capture_output=True, text=True, timeout=300)
    finally:
        os.remove(script)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l) for l in out.stderr.splitlines() if l.startswith("{")]
    assert len(recs) == 2, out.stderr[-2000:]
    by = {r["rank"]: r for r in recs}
    assert by[0]["got"].startswith("skipped: setup failed on another rank")
    assert by[1]["got"].startswith("skipped: MemoryError")
    assert all(r["next"] == 42 and r["sum"] == 2.0 for r in recs)


def test_no_roofline_fraction_can_exceed_one():
    """the bf16x3 training records printed frac 1.9 (algorithmic FLOPs against the fp32 MFMA peak while the kernels run three
    bf16 MFMAs per product).  They are priced on the pipe they use now, and compact_line refuses any `frac*` above 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    f = B.mfma_frac_fields("bf16x3", 308.9)                     # the round-5 step: 308.9 TF algorithmic
    assert abs(f["frac_of_mfma_peak"] - 3 * 308.9 / 2500.0) < 1e-12 and f["frac_of_mfma_peak"] < 1
    assert f["algorithmic_tflops"] == 308.9 and abs(f["x_fp32_mfma_peak"] - 308.9 / 157.3) < 1e-12
    assert B.mfma_frac_fields("fp32", 130.0)["frac_of_mfma_peak"] == 130.0 / 157.3
    res = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.9}, "records": {"x": {"roofline": {"frac": 1.96}}}}
    assert B.fracs_above_one(res) == [("/records/x/roofline/frac", 1.96)]
    try:
        B.compact_line(res)
    except AssertionError:
        pass
    else:
        raise AssertionError("compact_line accepted a fraction above 1")
