"""infer/sec of `simple` and `simple_string` at concurrency 16 for every tensor transport (wire,
system shm, device shm); TCA_HIPSHM_FAKE=1 runs it without a GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
GPU = None if os.environ.get("TCA_HIPSHM_FAKE") == "1" else 0
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server
with spawn_server(models=["simple", "simple_string"], gpu=GPU) as srv:
    common = ["-u", srv.grpc_url, "-i", "grpc", "--native", "--concurrency-range", "16", "-p", "800",
              "--stability-windows", "2", "--warmup-ms", "300"]
    for model in ("simple", "simple_string"):
        for shm in ("none", "system", "hip"):
            rows = perf(["-m", model, "--shared-memory", shm] + common)
            r = rows[0]
            print("RESULT", model, shm, round(r["infer_per_sec"]), "p50", round(r["p50_us"]), "server us: queue", round(r["server_queue_us"]), "in", round(r["server_compute_input_us"]), "infer", round(r["server_compute_infer_us"]), "out", round(r["server_compute_output_us"]), flush=True)
