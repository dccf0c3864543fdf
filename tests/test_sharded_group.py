"""The sharded execution group (one batch split over several replicas): data plane through
peer-mapped buffers and device-polled flags, or point-to-point collectives; no control
traffic and no process-group call per step in direct mode."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "_shard_driver.py")


def _run(extra, env_extra, timeout=420):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-u", DRIVER, *extra], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("SHARD_OK ")]
    assert p.returncode == 0 and lines, f"rc={p.returncode}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    return json.loads(lines[-1][len("SHARD_OK "):])


@pytest.mark.parametrize("world,mode", [(2, "direct"), (3, "direct"), (3, "rccl")])
def test_sharded_group_host_backed(world, mode):
    """Host memory stands in for the GPUs; mode "rccl" runs its point-to-point operations on gloo."""
    info = _run(["--world", str(world), "--mode", mode, "--steps", "12"], {"TCA_HIPSHM_FAKE": "1"})
    assert info["control_messages"] == 0 and info["process_group_calls"] == 0


@pytest.mark.parametrize("world", [2, 3, 4])
def test_collective_gather_brings_every_shard_home(world):
    """Mode "rccl" with one all_gather_into_tensor per step for the output rows (gloo here): steps of 32, 8,
    5, 1, 2 and 17 rows cover even and uneven shards and steps smaller than the group."""
    info = _run(["--world", str(world), "--mode", "rccl", "--gather", "collective", "--steps", "12"],
                {"TCA_HIPSHM_FAKE": "1"})
    assert info["steps"] == 12


@pytest.mark.parametrize("mode", ["direct", "rccl"])
def test_sequence_numbers_cross_the_32_bit_wrap(mode):
    """Mailbox words, flags and the progress word hold 32 bits of an unbounded step count."""
    info = _run(["--world", "3", "--mode", mode, "--steps", "24", "--first-seq", str(2 ** 32 - 9)],
                {"TCA_HIPSHM_FAKE": "1"})
    assert info["steps"] == 24


def test_next_seq_skips_the_empty_word():
    from triton_client_amd.parallel.sharded import RING, _behind, next_seq

    assert next_seq(0) == 1 and next_seq(2 ** 32 - 2) == 2 ** 32 - 1 and next_seq(2 ** 32 - 1) == 2 ** 32 + 1
    assert _behind(2 ** 32 + 3, 0xFFFFFFFE) == 5 and _behind(7, 7) == 0 and _behind(7, 9) == 0
    assert _behind(2 ** 32 + 1 + RING, 1) == RING


def test_worker_death_is_reported_not_waited_out():
    info = _run(["--world", "3", "--mode", "direct", "--kill-worker-after", "5"], {"TCA_HIPSHM_FAKE": "1"}, timeout=120)
    assert info["error"] and "worker 1" in info["error"], info
    assert info["seconds_to_error"] < 6 and info["seconds_to_close"] < 30, info


def test_worker_that_never_starts_fails_the_setup_at_once():
    """A replica process that exits during start-up ends accept_workers() with its exit code."""
    import time

    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['TCA_HIPSHM_FAKE'] = '1'\n"
        "from triton_client_amd.parallel.sharded import ShardRoot, ShardError\n"
        "root = ShardRoot(lambda x, o, s: None, 2, 0, 12, 16, 32)\n"
        "t0 = time.monotonic()\n"
        "try:\n"
        "    root.spawn_workers([0], model='no_such_model', startup_timeout=120.0)\n"
        "    print('NO_ERROR')\n"
        "except ShardError as e:\n"
        "    print('SHARD_ERROR', round(time.monotonic() - t0, 1), e)\n"
        "root.close()\n" % ROOT)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110,
                       env=dict(os.environ, TCA_HIPSHM_FAKE="1"))
    assert "SHARD_ERROR" in p.stdout, p.stdout + p.stderr[-2000:]
    assert "exited with code" in p.stdout or "failed during setup" in p.stdout, p.stdout
    assert time.monotonic() - t0 < 60


def test_server_outlives_a_dead_replica():
    """Requests for the sharded model fail with the worker's name within the group's time limit; other
    models, health and shutdown are unaffected."""
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "_shard_server_driver.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, TCA_HIPSHM_FAKE="1"))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("SHARD_SERVER_OK ")]
    assert p.returncode == 0 and lines, f"rc={p.returncode}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    report = json.loads(lines[-1][len("SHARD_SERVER_OK "):])
    assert all(e and "shard worker 2" in e for e in report["errors"]), report
    assert report["seconds_for_three_failed_requests"] < 10 and report["seconds_to_stop"] < 30, report
    assert report["live"] and report["ready"] and report["other_model_ready"], report


@pytest.mark.gpu
@pytest.mark.parametrize("flags", ["kernel", "stream"])
def test_sharded_group_hip_ipc_one_gpu(flags):
    """Real hipIpc mappings, peer stores and flags polled by the GPU: three ranks share GPU 0."""
    info = _run(["--world", "3", "--mode", "direct", "--device", "0", "--steps", "12", "--flags", flags], {})
    assert info["control_messages"] == 0 and info["process_group_calls"] == 0


@pytest.mark.gpu
def test_fanout_bench_rehearsal_on_one_gpu():
    """``tools/fanout_bench.py`` with three replica processes on GPU 0: peer-access matrix, step times and
    bytes per second into every peer's inbox."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fanout_bench.py"), "--devices", "0,0", "--batch", "96",
                        "--iters", "40", "--modes", "direct"], cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines[0]["peer_access"] == {"gpus": [0], "matrix": [[1]]}
    run = lines[1]
    assert run["mode"] == "direct" and run["replicas"] == 3 and run["same_gpu"] is True
    assert len(run["peer_GBps"]) == 2 and all(v > 1 for v in run["peer_GBps"]) and run["all_peers_GBps"] > 1
    assert run["us_per_step_pipelined"] > 0


@pytest.mark.gpu
@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="needs two GPUs")
@pytest.mark.parametrize("mode,gather", [("direct", None), ("rccl", "p2p"), ("rccl", "collective")])
def test_sharded_group_across_two_gpus(mode, gather):
    """The cross-GPU data plane proper: rank 1 on GPU 1 (xGMI peer stores / RCCL)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fanout_bench.py"), "--devices", "1", "--batch", "64",
                        "--iters", "40", "--modes", mode] + (["--gather", gather] if gather else []), cwd=ROOT,
                       capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines[1]["mode"] == mode and lines[1]["same_gpu"] is False and lines[1]["us_per_step_pipelined"] > 0
