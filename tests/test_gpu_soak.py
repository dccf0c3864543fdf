"""A serving process must not grow while it serves: requests whose tensors sit in system shared memory go through
the native executor's staging copies (host-to-device ``hipMemcpyAsync`` on a copy stream), and the HIP runtime keeps
about a kilobyte per such copy until the stream is synchronised - 11 MB per second at this load when the copy stream
was only ever waited on through events."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOADGEN = os.path.join(ROOT, "triton_client_amd", "_lib", "tca_loadgen")


@pytest.mark.gpu
def test_resident_set_stays_level_under_staged_load():
    import psutil
    import tritonclient.grpc as grpcclient
    import tritonclient.utils.shared_memory as sysshm

    from triton_client_amd.server import spawn_server

    c = 32
    with spawn_server(models=["resnet50"], gpu=0) as srv:
        server = psutil.Process(srv.proc.pid)
        ctl = grpcclient.InferenceServerClient(srv.grpc_url)
        x = np.random.default_rng(0).standard_normal((8, 3, 224, 224)).astype(np.float16)
        regions = []
        try:
            for i in range(c):
                hin = sysshm.create_shared_memory_region(f"soak_in{i}", f"/tca_soaktest_{os.getpid()}_in{i}", x.nbytes)
                hout = sysshm.create_shared_memory_region(f"soak_out{i}", f"/tca_soaktest_{os.getpid()}_out{i}", 32000)
                regions += [hin, hout]
                sysshm.set_shared_memory_region(hin, [x])
                ctl.register_system_shared_memory(f"soak_in{i}", f"/tca_soaktest_{os.getpid()}_in{i}", x.nbytes)
                ctl.register_system_shared_memory(f"soak_out{i}", f"/tca_soaktest_{os.getpid()}_out{i}", 32000)
            cmd = [LOADGEN, "-u", srv.grpc_url, "-i", "grpc", "-m", "resnet50", "-c", str(c), "-b", "8", "--connections", "8",
                   "-d", "5", "--warmup", "0.3", "--shm-input", f"input:FP16:8,3,224,224:soak_in{{slot}}:{x.nbytes}",
                   "--shm-output", "output:soak_out{slot}:32000"]
            rss, rates = [], []
            for _ in range(3):
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
                r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                assert r["errors"] == 0, r
                rates.append(r["infer_per_sec"])
                rss.append(server.memory_info().rss / 2 ** 20)
            # the first window may still touch fresh pages (arena, pools); the next two must be level
            assert rss[2] - rss[1] < 25 and rss[1] - rss[0] < 40, (rss, rates)
            assert min(rates) > 20000, rates
        finally:
            try:
                ctl.unregister_system_shared_memory()
            finally:
                for h in regions:
                    sysshm.destroy_shared_memory_region(h)
                ctl.close()
