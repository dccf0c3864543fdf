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


@pytest.mark.gpu
@pytest.mark.parametrize("flags", ["kernel", "stream"])
def test_sharded_group_hip_ipc_one_gpu(flags):
    """Real hipIpc mappings, peer stores and flags polled by the GPU: three ranks share GPU 0."""
    info = _run(["--world", "3", "--mode", "direct", "--device", "0", "--steps", "12", "--flags", flags], {})
    assert info["control_messages"] == 0 and info["process_group_calls"] == 0
