#!/usr/bin/env python3
"""Does reading hipIpc-mapped memory cost the rest of the GPU something?  The parent allocates 16 regions of 2.4 MB
(what 16 batch-8 requests name) and exports them; the child opens them, loads ResNet-50, and times batch-128 forwards
on one stream while a second stream gathers 16 x 2.4 MB every 1.6 ms - from the IPC mappings, or from its own memory."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
N, BYTES = 16, 8 * 3 * 224 * 224 * 2


def child(handles):
    import torch

    from triton_client_amd.models import MODEL_ZOO
    from triton_client_amd.ops.copy import batched_copy
    from triton_client_amd.utils import hip_shared_memory as hipshm

    maps = [hipshm.open_shared_memory_region(f"probe{i}", bytes.fromhex(h), BYTES, 0) for i, h in enumerate(handles)]
    model = MODEL_ZOO["resnet50"]()
    model.load(0)
    x = torch.randn(128, 3, 224, 224, device="cuda").half()
    own = torch.empty(N * BYTES, dtype=torch.uint8, device="cuda")
    dst = torch.empty(N * BYTES, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    for _ in range(5):
        model.run(x)
    model.stream.synchronize()
    sources = {"no gather": None,
               "gather from own memory": [(own.data_ptr() + i * BYTES, dst.data_ptr() + i * BYTES, BYTES) for i in range(N)],
               "gather from IPC mappings": [(maps[i].address, dst.data_ptr() + i * BYTES, BYTES) for i in range(N)]}
    for rep in range(2):
        for name, triples in sources.items():
            if triples is not None:
                t0 = time.perf_counter()
                for _ in range(50):
                    batched_copy(triples, side)
                side.synchronize()
                alone = (time.perf_counter() - t0) / 50 * 1e6
            else:
                alone = 0.0
            model.stream.synchronize()
            t0 = time.perf_counter()
            iters = 200
            for _ in range(iters):
                model.run(x)
                if triples is not None:
                    batched_copy(triples, side)  # one gather per forward, on the other stream
            model.stream.synchronize()
            side.synchronize()
            dt = (time.perf_counter() - t0) / iters
            print(f"{name:28s}: forward {dt * 1e3:.3f} ms ({128 / dt:.0f} img/s), the gather alone {alone:.1f} us", flush=True)
    model.unload()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(sys.argv[2:])
    from triton_client_amd.utils import hip_shared_memory as hipshm

    regions = [hipshm.create_shared_memory_region(f"probe_src{i}", BYTES, 0) for i in range(N)]
    handles = [bytes(r._raw).hex() for r in regions]
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "child", *handles])
    for r in regions:
        hipshm.destroy_shared_memory_region(r)
    sys.exit(rc)


if __name__ == "__main__":
    main()
