"""ResNet-50 whose batches are split over several GPUs.

``resnet50_sharded`` is the servable form of :mod:`triton_client_amd.parallel.sharded`: the
serving process owns the ingest GPU (rank 0) and starts one replica process per further GPU.
A request (or the batch the dynamic batcher merged) is split along the batch axis, the shards
travel over xGMI, and the logits of every shard land in the caller's output buffer - a
registered device shared-memory region, typically.  The reference has nothing to compare with
(a single-process client); the contract is the one of the plain classifier
(``src/python/examples/image_client.py:59-150``): one FP16 NCHW input, one FP32 vector output.

Environment:

``TCA_SHARD_DEVICES``  comma separated GPU ordinals of ranks 1..N-1 (default: every visible GPU
                       other than the serving one; the same ordinal may repeat, which puts
                       several replicas on one GPU - used by tests on a one-GPU machine)
``TCA_SHARD_MODE``     ``direct`` (peer stores + flags, default) or ``rccl``
``TCA_SHARD_FLAGS``    ``kernel`` (default) or ``stream`` (command-processor wait packets)
"""

from __future__ import annotations

import os

from .base import ModelError, TensorSpec, register
from .vision import ResNet50


@register
class ResNet50Sharded(ResNet50):
    name = "resnet50_sharded"
    native_kernel = None  # one forward = a step of the whole replica group, driven from Python
    autoload = False  # starts replica processes: only loaded when asked for by name
    inputs = [TensorSpec("input", "FP16", [3, 224, 224], format="FORMAT_NCHW")]
    outputs = [TensorSpec("output", "FP32", [1000], label_filename="labels.txt")]

    def __init__(self):
        super().__init__()
        self._shard = None
        self.worker_devices = None
        spec = os.environ.get("TCA_SHARD_DEVICES")
        if spec is not None:
            self.worker_devices = [int(v) for v in spec.split(",") if v.strip() != ""]
        self.shard_mode = os.environ.get("TCA_SHARD_MODE", "direct")
        self.per_gpu_batch = int(os.environ.get("TCA_SHARD_PER_GPU_BATCH", "128"))
        self._configure(1 + len(self.worker_devices) if self.worker_devices is not None else 1)

    def _configure(self, world):
        self.world = world
        self.max_batch_size = self.per_gpu_batch * world
        # the batcher merges requests of 8: every merged total whose shards are graphs the
        # replicas captured at start-up (multiples of 8 per GPU)
        step = 8 * world
        self.warmup_batches = tuple(range(step, self.max_batch_size + 1, step))

    def load(self, device_index=None):
        import torch

        index = device_index or 0
        if self.worker_devices is None:
            self.worker_devices = [d for d in range(torch.cuda.device_count()) if d != index]
            self._configure(1 + len(self.worker_devices))
        super().load(device_index)
        if self.world == 1:
            return
        from ..parallel.sharded import SLOTS, ShardRoot

        if self.instance_count != SLOTS:
            raise ModelError(f"resnet50_sharded needs {SLOTS} execution instances (TCA_RESNET_INSTANCES)")
        shard_rows = -(-self.max_batch_size // self.world)
        self._shard = ShardRoot(self._forward_shard, self.world, index, 3 * 224 * 224 * 2, 1000 * 4,
                                self.max_batch_size, streams=self._streams, mode=self.shard_mode)
        warm = sorted({b // self.world for b in self.warmup_batches} | {shard_rows})
        try:
            self._shard.spawn_workers(self.worker_devices, model="resnet50", warm_rows=warm)
        except Exception:
            self._shard.close()
            self._shard = None
            raise

    def _forward_shard(self, x, out, slot):
        ResNet50._run_stable(self, x, out, slot)

    def _run_stable(self, x, out, instance):
        if self._shard is None:
            return super()._run_stable(x, out, instance)
        self._shard.step(x, out, instance)
        return out

    def unload(self):
        shard, self._shard = self._shard, None
        if shard is not None:
            shard.close()
        super().unload()
