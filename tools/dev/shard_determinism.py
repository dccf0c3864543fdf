"""Diagnostic: are the plain and the sharded ResNet-50 bit-reproducible, and do they agree?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_client_amd.server import spawn_server  # noqa: E402

srv = spawn_server(models=["resnet50", "resnet50_sharded"], gpu=0,
                   env={"TCA_SHARD_DEVICES": "0,0", "TCA_SHARD_PER_GPU_BATCH": "32", "TCA_RESNET_MAX_BATCH": "32"},
                   startup_timeout=600)
import tritonclient.grpc as g  # noqa: E402
import tritonclient.utils.cuda_shared_memory as hipshm  # noqa: E402

batch = 48
x = np.random.default_rng(1).standard_normal((batch, 3, 224, 224)).astype(np.float16)
try:
    with g.InferenceServerClient(srv.grpc_url) as c:
        def plain(rows):
            r = c.infer("resnet50", [g.InferInput("input", list(rows.shape), "FP16").set_data_from_numpy(rows)])
            return r.as_numpy("output").copy()

        a = [plain(x[:16]) for _ in range(3)]
        print("plain b16 repeat max diff", [float(np.abs(a[0] - v).max()) for v in a[1:]])
        b = [plain(x[16:32]) for _ in range(2)]
        print("plain rows16-32 repeat", float(np.abs(b[0] - b[1]).max()))
        in_bytes, out_bytes = batch * 3 * 224 * 224 * 2, batch * 1000 * 4
        hin = hipshm.create_shared_memory_region("sh_in", in_bytes, 0)
        hout = hipshm.create_shared_memory_region("sh_out", out_bytes, 0)
        hipshm.set_shared_memory_region(hin, [x])
        c.register_cuda_shared_memory("sh_in", hipshm.get_raw_handle(hin), 0, in_bytes)
        c.register_cuda_shared_memory("sh_out", hipshm.get_raw_handle(hout), 0, out_bytes)
        inp = g.InferInput("input", [batch, 3, 224, 224], "FP16")
        inp.set_shared_memory("sh_in", in_bytes)
        out = g.InferRequestedOutput("output")
        out.set_shared_memory("sh_out", out_bytes)
        s = []
        for _ in range(4):
            c.infer("resnet50_sharded", [inp], outputs=[out])
            s.append(hipshm.get_contents_as_numpy(hout, np.float32, [batch, 1000]).copy())
        print("sharded repeat max diff", [float(np.abs(s[0] - v).max()) for v in s[1:]])
        ref = np.concatenate([a[0], b[0], plain(x[32:48])])
        for r in range(3):
            print("rank", r, "rows vs plain: max diff",
                  float(np.abs(s[0][16 * r:16 * r + 16] - ref[16 * r:16 * r + 16]).max()),
                  "later run", float(np.abs(s[3][16 * r:16 * r + 16] - ref[16 * r:16 * r + 16]).max()))
        # the same 16 rows through a registered region with the plain model (graph on the region)
        inp16 = g.InferInput("input", [16, 3, 224, 224], "FP16")
        inp16.set_shared_memory("sh_in", 16 * 3 * 224 * 224 * 2)
        out16 = g.InferRequestedOutput("output")
        out16.set_shared_memory("sh_out", 16 * 1000 * 4)
        c.infer("resnet50", [inp16], outputs=[out16])
        z = hipshm.get_contents_as_numpy(hout, np.float32, [16, 1000]).copy()
        print("plain via region vs plain via wire", float(np.abs(z - a[0]).max()))
        c.unregister_cuda_shared_memory()
finally:
    srv.stop()
