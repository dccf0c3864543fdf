"""Data-parallel execution of one batched request across model replicas.

Every rank runs the same loop: the ingest rank announces the next batch (a
tiny control broadcast carrying the batch size, or -1 to stop), the input is
fanned out with :class:`BatchFanout`, each rank runs its local replica on its
shard and the logits are all-gathered.  With RCCL the two collectives of a
step are enqueued on the same HIP stream as the model, so there is no host
synchronisation between scatter, forward and gather.
"""

from __future__ import annotations

from .fanout import BatchFanout


class ShardedModelGroup:
    """Parameters
    ----------
    forward : callable(tensor) -> tensor
        The local replica (e.g. ``ResNet50.run``).
    sample_shape : tuple
        Shape of one input sample (without batch axis).
    in_dtype, device : torch dtype / device of the inputs on this rank
    group, root, mode : see :class:`BatchFanout`; ``mode="direct"`` moves the data
        through peer-mapped buffers (:class:`DirectPeerExchange`) instead of
        collectives - then only the root gets the outputs, the other ranks get an
        empty tensor - and needs ``max_batch`` and ``out_row_bytes`` (bytes of one
        output row) to size the buffers.
    """

    def __init__(self, forward, sample_shape, in_dtype, device, group=None, root=0, mode="scatter", max_batch=256,
                 out_row_bytes=4096):
        self.forward = forward
        self.sample_shape = tuple(sample_shape)
        self.in_dtype = in_dtype
        self.device = device
        self.fan = BatchFanout(group, root, "scatter" if mode == "direct" else mode)
        self.rank = self.fan.rank
        self.root = root
        self.direct = None
        if mode == "direct":
            import math

            import torch

            from .direct import DirectPeerExchange

            rows = -(-max_batch // self.fan.world)
            sample_bytes = math.prod(self.sample_shape) * torch.empty(0, dtype=in_dtype).element_size()
            index = torch.device(device).index or 0
            self.direct = DirectPeerExchange(rows * sample_bytes, rows * out_row_bytes, device=index, group=group,
                                             root=root)

    def _announce(self, batch):
        import torch

        flag = torch.tensor([batch if batch is not None else 0], dtype=torch.int64, device=self.device)
        self.fan.dist.broadcast(flag, src=self.fan._global(self.root), group=self.fan.group)
        return int(flag.item())

    def step(self, full=None):
        """One sharded inference.  ``full`` is the whole batch on the root
        (ignored elsewhere).  Returns the gathered outputs on every rank, or
        ``None`` when the root signalled shutdown with :meth:`stop`."""
        batch = self._announce(full.shape[0] if (self.rank == self.root and full is not None) else None)
        if batch < 0:
            return None
        if self.direct is not None:
            shard = self.direct.scatter(full, (batch, *self.sample_shape), self.in_dtype)
            local = self.forward(shard)
            out = self.direct.gather(local, batch)
            return out if out is not None else local[:0]
        shard = self.fan.fan_out(full, (batch, *self.sample_shape), self.in_dtype, self.device)
        local = self.forward(shard)
        return self.fan.gather(local, batch)

    def stop(self):
        """Root only: release the other ranks from :meth:`serve`."""
        import torch

        flag = torch.tensor([-1], dtype=torch.int64, device=self.device)
        self.fan.dist.broadcast(flag, src=self.fan._global(self.root), group=self.fan.group)

    def close(self):
        if self.direct is not None:
            self.direct.close()
            self.direct = None

    def serve(self):
        """Non-root ranks: participate in steps until the root calls :meth:`stop`."""
        while self.step() is not None:
            pass
