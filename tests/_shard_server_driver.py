"""A server whose model fans its batches out over replica processes keeps serving when a replica dies:
requests for that model get an error that names the worker (within the group's time limit, no hang),
every other model and the health endpoints answer as before.  Host memory stands in for the GPUs
(``TCA_HIPSHM_FAKE=1``, set before the device shared-memory library loads, hence a process of its own).
Prints ``SHARD_SERVER_OK <json>``."""
import json
import os
import sys
import time

os.environ["TCA_HIPSHM_FAKE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    import torch

    import triton_client_amd.http as httpclient
    from triton_client_amd.models.base import Model, TensorSpec, register
    from triton_client_amd.parallel.sharded import ShardRoot
    from triton_client_amd.server import Server
    from triton_client_amd.utils import InferenceServerException

    @register
    class RowSumSharded(Model):
        """out[i, :] = sum(x[i]) + 1000 * rank of the replica that took row i (three replicas)."""

        name = "rowsum_sharded"
        max_batch_size = 32
        inputs = [TensorSpec("input", "FP16", [2, 3])]
        outputs = [TensorSpec("output", "FP32", [4])]

        def load(self, device_index=None):
            def forward(x, out, slot):
                out.copy_(x.float().reshape(x.shape[0], -1).sum(dim=1, keepdim=True).expand_as(out))

            self.shard = ShardRoot(forward, 3, 0, 6 * 2, 4 * 4, 32, timeout_ms=3000)
            self.shard.spawn_workers([0, 0], model="test_rowsum", sample_shape=(2, 3), out_shape=(4,),
                                     startup_timeout=300.0)

        def unload(self):
            self.shard.close()

        def execute(self, inputs, ctx):
            x = torch.from_numpy(np.array(inputs["input"]))
            out = torch.zeros(x.shape[0], 4, dtype=torch.float32)
            self.shard.step(x, out, 0)
            self.shard.check()
            return {"output": out.numpy()}

    server = Server(models=["rowsum_sharded", "simple"])
    server.start()
    report = {}
    try:
        client = httpclient.InferenceServerClient(server.http_url)

        def sharded(rows):
            x = (np.arange(rows * 6, dtype=np.float32).reshape(rows, 2, 3) % 7).astype(np.float16)
            inp = httpclient.InferInput("input", list(x.shape), "FP16")
            inp.set_data_from_numpy(x)
            got = client.infer("rowsum_sharded", [inp]).as_numpy("output")
            base = x.astype(np.float32).reshape(rows, -1).sum(axis=1)
            sizes = [rows // 3 + (1 if r < rows % 3 else 0) for r in range(3)]
            rank = np.repeat(np.arange(3), sizes)
            assert np.array_equal(got[:, 0], base + 1000 * rank), (got[:, 0], base + 1000 * rank)

        def simple():
            a = np.arange(16, dtype=np.int32).reshape(1, 16)
            i0, i1 = httpclient.InferInput("INPUT0", [1, 16], "INT32"), httpclient.InferInput("INPUT1", [1, 16], "INT32")
            i0.set_data_from_numpy(a)
            i1.set_data_from_numpy(a)
            r = client.infer("simple", [i0, i1])
            assert np.array_equal(r.as_numpy("OUTPUT0"), a + a)

        for rows in (9, 32, 4):
            sharded(rows)
        simple()
        model = server.core._get("rowsum_sharded").model
        model.shard._procs[1].kill()
        errors = []
        t0 = time.monotonic()
        for _ in range(3):
            try:
                sharded(12)
                errors.append(None)
            except InferenceServerException as e:
                errors.append(str(e))
        report["seconds_for_three_failed_requests"] = round(time.monotonic() - t0, 2)
        report["errors"] = errors
        simple()
        report["live"] = client.is_server_live()
        report["ready"] = client.is_server_ready()
        report["other_model_ready"] = client.is_model_ready("simple")
        client.close()
    finally:
        t0 = time.monotonic()
        server.stop()
        report["seconds_to_stop"] = round(time.monotonic() - t0, 2)
    print("SHARD_SERVER_OK", json.dumps(report))


if __name__ == "__main__":
    main()
