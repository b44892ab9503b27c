"""bench.py's start-up logic: who brings the ranks up (no GPU needed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench


def test_one_gpu_or_a_started_rank_needs_no_launcher():
    assert bench.launcher_command(1, "torch", {}, []) is None
    assert bench.launcher_command(8, "torch", {"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]) is None   # a rank already
    assert bench.launcher_command(8, "group", {}, ["--gpus", "8", "--driver", "group"]) is None           # in-process ranks


def test_plain_start_with_several_gpus_launches_the_ranks():
    argv = ["--gpus", "4", "--steps", "5", "--backend", "gloo"]
    cmd = bench.launcher_command(4, "torch", {}, argv, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == argv                     # the ranks get the same arguments
    port = bench.free_port()
    assert 1024 < port < 65536


def test_arguments():
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.driver, a.backend) == (1, 8, 3, "torch", "nccl")
    a = bench.parse_args(["--gpus", "8", "--driver", "group", "--share-gpus"])
    assert a.gpus == 8 and a.driver == "group" and a.share_gpus


def test_ranks_are_started_and_their_failure_is_passed_on():
    """Without a GPU every rank stops at once with the same message: the parent has started two of them (their
    output says so) and passes their failure on as its own exit status."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                          "--warmup", "0", "--jobs", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs a GPU") >= 2, res.stderr[-2000:]


def test_statistics_are_summed_over_ranks_and_timed_by_the_slowest():
    a = {k: 1 for k in bench.STAT_KEYS}
    a.update({k: 2.0 for k in bench.TIME_KEYS})
    b = {k: 3 for k in bench.STAT_KEYS}
    b.update({k: 5.0 for k in bench.TIME_KEYS})
    vals, times = bench.sum_stats([a, b], 2)
    assert all(vals[k] == 2.0 for k in bench.STAT_KEYS) and all(times[k] == 2.5 for k in bench.TIME_KEYS)
    acc = {}
    bench.add_counted(acc, {"node_visits_camera": 10, "prim_tests_camera": 4, "node_visits_light": 6, "prim_tests_light": 2,
                            "rays": 5, "rays_camera": 3}, 3)
    vals, _ = bench.sum_stats([acc], 3)
    assert vals["node_visits_camera"] == 10 and vals["counted_rays"] == 5 and vals["counted_rays_camera"] == 3
