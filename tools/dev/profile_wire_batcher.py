#!/usr/bin/env python3
"""cProfile of the batcher / completer threads of a server under wire load (concurrency 16)."""
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
out = os.path.abspath("gpurun_out/batcher_prof")
os.environ["TCA_PROFILE_DIR"] = out
from triton_client_amd.perf import main as perf
from triton_client_amd.server import spawn_server

with spawn_server(models=["resnet50"], gpu=0) as srv:
    rows = perf(["-m", "resnet50", "-u", srv.grpc_url, "-i", "grpc", "-b", "8", "--native", "--shared-memory",
                 os.environ.get("TCA_PROF_SHM", "none"),
                 "--concurrency-range", os.environ.get("TCA_PROF_CONC", "64"), "-p", "2000", "--stability-windows",
                 "2", "--warmup-ms", "400"])
    print("RESULT", round(rows[0]["infer_per_sec"]), rows[0]["requests"])
for name in sorted(os.listdir(out)):
    print("=====", name)
    st = pstats.Stats(os.path.join(out, name))
    st.sort_stats("tottime").print_stats(int(os.environ.get("TCA_PROF_ROWS", "14")))
