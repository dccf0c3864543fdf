"""bench.py prints the one JSON line its callers parse, single rank and under
``torch.distributed.run`` (two ranks sharing the one GPU of a test box, over gloo: RCCL
refuses two ranks on one device; with one GPU per rank the default backend is RCCL)."""

import json
import os
import subprocess
import sys

import pytest

# generous limits: on a fresh machine the first `import torch` of a child process alone can take minutes
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(420)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _check(line, gpus, steps, warmup):
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == gpus and line["steps"] == steps and line["warmup"] == warmup
    assert line["unit"] == "infer/sec" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["dtype"] == "fp16" and line["config"]["model"] == "resnet50" and line["config"]["batch"] == 8
    assert line["value"] > 0 and line["ms_per_step"] > 0
    # value is the whole job: requests x batch over the timed wall clock
    per_step = line["config"]["requests_per_step_per_gpu"] * 8 * gpus
    assert line["value"] == pytest.approx(per_step / (line["ms_per_step"] / 1e3), rel=1e-3)
    assert line["p50_latency_us"] <= line["p99_latency_us"]
    assert line["ranks"] == gpus and len(line["per_gpu_infer_per_sec"]) == gpus
    # the shared-memory-versus-wire comparison is part of the same run
    for key in ("wire", "system_shm"):
        assert line["extra"][key]["infer_per_sec"] > 0
        assert line["extra"][key]["requests"] == 128 * gpus
    # the concurrency sweep over HIP shared memory and the content check against a direct forward
    sweep = line["extra"]["sweep"]
    assert [row["concurrency_per_gpu"] for row in sweep] == [1, 16]
    for row in sweep:
        assert row["infer_per_sec"] > 0 and row["p50_latency_us"] <= row["p99_latency_us"] and row["window_s"] > 0.2
    check = line["extra"]["check"]
    assert check["ok"] is True, check
    assert check["max_abs_err"] < 0.05 * check["logit_scale"] and check["min_distance_between_images"] > 4 * check["max_abs_err"]
    # ... and so are the other configurations of the benchmark set, summed over the ranks
    secondary = line["extra"]["secondary"]
    assert "error" not in secondary, secondary
    for key in ("simple_http_sysshm", "simple_grpc_wire", "string_grpc_hipshm"):
        assert secondary[key]["infer_per_sec"] > 0 and secondary[key]["p50_latency_us"] <= secondary[key]["p99_latency_us"]
    assert secondary["decoupled_stream"]["responses_per_sec"] > 0


def test_single_rank_line():
    cmd = [sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--requests-per-step", "64", "--concurrency",
           "16",
           "--extra-requests", "128", "--sweep", "1,16", "--sweep-seconds", "0.4", "--image-requests", "256",
           "--secondary-window-ms", "400"]
    done = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert done.returncode == 0, done.stderr[-2000:]
    _check(_json_line(done.stdout), 1, 3, 1)


def test_two_ranks_print_one_aggregate_line():
    env = dict(os.environ, TCA_BENCH_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--requests-per-step", "64", "--concurrency", "16",
           "--extra-requests", "128", "--sweep", "1,16", "--sweep-seconds", "0.4", "--image-requests", "256",
           "--secondary-window-ms", "400"]
    done = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert done.returncode == 0, done.stderr[-2000:]
    _check(_json_line(done.stdout), 2, 3, 1)


def test_fanout_is_opt_in_and_checks_its_answers():
    """``--mode fanout`` on a one-GPU box: three replicas share the GPU (rehearsal of the cross-GPU form);
    the logits the group served are compared with a single-GPU forward in the same run."""
    cmd = [sys.executable, "bench.py", "--mode", "fanout", "--steps", "2", "--warmup", "1", "--requests-per-step", "64",
           "--concurrency", "16", "--extra-requests", "128", "--fanout-requests", "48", "--fanout-modes", "direct"]
    done = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert done.returncode == 0, done.stderr[-2000:]
    fan = _json_line(done.stdout)["extra"]["fanout"]
    assert fan["replicas"] == 3 and "parity unpinned on >1 GPU" in fan["parity"]
    assert "error" not in fan["direct"], fan
    assert fan["direct"]["infer_per_sec"] > 0 and fan["direct"]["check"]["ok"] is True, fan


def test_gpus_flag_alone_starts_the_ranks():
    """``python bench.py --gpus 2`` with no launcher: bench.py starts both ranks itself."""
    env = dict(os.environ, TCA_BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--requests-per-step", "64",
           "--concurrency", "16", "--extra-requests", "128", "--sweep", "1,16", "--sweep-seconds", "0.4",
           "--image-requests", "256", "--secondary-window-ms", "400"]
    done = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert done.returncode == 0, done.stderr[-2000:]
    line = _json_line(done.stdout)
    _check(line, 2, 3, 1)
    assert line["rank_backend"] == "gloo"
    # the ranks proved themselves before measuring (on one GPU: gloo, so not "RCCL verified"); fan-out is opt-in
    pre = line["preflight"]
    assert pre["ranks_verified"] is True and pre["rccl_ranks_verified"] is False and pre["ranks"] == 2
    assert pre["peer_access"] == [[1]] and "parity unpinned on >1 GPU" in line["extra"]["fanout"]["skipped"]


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="needs two GPUs")
def test_two_ranks_on_two_gpus_over_rccl():
    """One rank per GPU on the RCCL backend, with the fan-out measured across the two GPUs."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TCA_BENCH_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--mode", "fanout", "--steps", "3", "--warmup", "1",
           "--requests-per-step", "64", "--concurrency", "16", "--extra-requests", "128", "--fanout-requests", "64"]
    done = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert done.returncode == 0, done.stderr[-2000:]
    line = _json_line(done.stdout)
    assert line["n_gpus"] == 2 and line["rank_backend"] == "rccl"
    assert line["preflight"]["rccl_ranks_verified"] is True and line["preflight"]["distinct_gpus"] == 2
    fan = line["extra"]["fanout"]
    for mode in ("direct", "rccl"):
        assert "error" not in fan[mode], fan
        assert fan[mode]["check"]["ok"] is True, fan


def test_rank_count_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    done = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, capture_output=True,
                          text=True, timeout=120, env=env)
    assert done.returncode != 0 and "one rank per GPU" in done.stderr
