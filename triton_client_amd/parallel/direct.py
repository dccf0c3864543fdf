"""Fan-out / gather through peer-mapped device buffers instead of collectives.

The reference has no multi-GPU data plane at all (its transports are HTTP,
gRPC, POSIX shm and CUDA IPC handles, ``src/python/library/tritonclient/utils/cuda_shared_memory/__init__.py:130-136``);
this is the xGMI-shaped counterpart of :class:`.fanout.BatchFanout`:

* every rank owns an **inbox** (a ``hipMalloc`` region exported with
  ``hipIpcGetMemHandle``); the ingest rank opens all of them once and a scatter
  is then N-1 plain device-to-device copies, each on its own HIP stream, i.e.
  one per point-to-point xGMI link, all in flight together - no ring, no
  staging, no per-call handshakes;
* the ingest rank owns a **gather buffer** every other rank has opened; a
  gather is one small copy per rank straight into its slot (outputs are a few
  KB and latency bound, so a single remote write beats a collective launch);
* control (buffer hand-over) is a barrier on the process group - gloo sockets
  or RCCL, whichever the group was created with.  Handles travel once, at
  construction, through ``all_gather_object``.

With ``TCA_HIPSHM_FAKE=1`` the regions are host memory and the same code runs
on CPUs (tests).  On one GPU several ranks may share the device: IPC handles
work between processes regardless of which GPUs they use.
"""

from __future__ import annotations

import os


class DirectPeerExchange:
    """Parameters
    ----------
    shard_bytes : capacity of every rank's inbox (largest input shard)
    out_bytes   : capacity of one rank's slot in the gather buffer
    device      : this rank's GPU ordinal
    group, root : process group (default group when None) and ingest rank
    """

    def __init__(self, shard_bytes, out_bytes, device=0, group=None, root=0, tag="tca_dp"):
        import torch.distributed as dist

        from ..utils import hip_shared_memory as hipshm

        self._hipshm = hipshm
        self.dist = dist
        self.group = group
        self.root = root
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = int(device)
        self.shard_bytes = int(shard_bytes)
        self.out_bytes = int(out_bytes)
        uniq = f"{tag}_{os.getpid()}_{self.rank}"
        self.inbox = hipshm.create_shared_memory_region(f"{uniq}_in", self.shard_bytes, self.device)
        self.gather_buf = None
        if self.rank == root:
            self.gather_buf = hipshm.create_shared_memory_region(f"{uniq}_out", self.out_bytes * self.world,
                                                                 self.device)
        mine = {"rank": self.rank, "device": self.device, "inbox": self.inbox._raw,  # the 64 raw handle bytes (get_raw_handle() is base64)
                "gather": self.gather_buf._raw if self.gather_buf is not None else None}
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        self._peers = {}      # rank -> opened inbox (root only)
        self._root_gather = None
        self._streams = {}
        self._fake = self.inbox.host_backed
        if self.rank == root:
            for info in everyone:
                if info["rank"] == root:
                    continue
                self._peers[info["rank"]] = hipshm.open_shared_memory_region(
                    f"{uniq}_peer{info['rank']}", info["inbox"], self.shard_bytes, info["device"])
                self._streams[info["rank"]] = self._new_stream()
        else:
            info = everyone[root]
            self._root_gather = hipshm.open_shared_memory_region(f"{uniq}_rootout", info["gather"],
                                                                 self.out_bytes * self.world, info["device"])
            self._streams[root] = self._new_stream()
        dist.barrier(group=group)

    # ------------------------------------------------------------------ helpers
    def _new_stream(self):
        if self._fake:
            return None
        from ..ops._device import Stream

        return Stream(self.device)

    def _push(self, region, offset, tensor, stream):
        """Enqueue ``tensor`` (contiguous, on this rank's device or host in fake mode)
        -> ``region[offset:]``."""
        from ..ops._device import check, lib

        nbytes = tensor.numel() * tensor.element_size()
        if nbytes > region.byte_size - offset:
            raise ValueError(f"{nbytes} bytes do not fit the peer buffer ({region.byte_size - offset} left)")
        if not tensor.is_contiguous():
            raise ValueError("tensors must be contiguous")
        src_device = -1 if self._fake else self.device
        if not self._fake:
            import torch

            torch.cuda.current_stream(tensor.device).synchronize()  # the producer's work comes first
        handle = stream.handle if stream is not None else None
        check(lib().tca_hipshm_copy_from_async(region._native(), int(offset), tensor.data_ptr(), nbytes, src_device,
                                               handle), "peer copy")

    def _sync(self):
        for s in self._streams.values():
            if s is not None:
                s.synchronize()

    def _view(self, region, dtype, shape, offset=0):
        """Tensor of torch ``dtype`` aliasing ``region[offset:]``."""
        import torch

        from ..utils import np_to_triton_dtype

        if dtype == torch.bfloat16:
            name = "BF16"
        else:
            name = np_to_triton_dtype(torch.empty(0, dtype=dtype).numpy().dtype)
        return self._hipshm.mapping_as_torch(region, name, list(shape), offset)

    # -------------------------------------------------------------------- API
    def shard_sizes(self, batch):
        base, extra = divmod(batch, self.world)
        return [base + (1 if r < extra else 0) for r in range(self.world)]

    def scatter(self, full, shape, dtype):
        """Distribute the rows of ``full`` (root only; ``shape`` = full shape, known
        to every rank) and return this rank's shard as a tensor aliasing its inbox."""
        sizes = self.shard_sizes(shape[0])
        tail = tuple(shape[1:])
        if self.rank == self.root:
            row = 0
            for r, n in enumerate(sizes):
                piece = full[row: row + n]
                row += n
                if n == 0:
                    continue
                if r == self.root:
                    self._view(self.inbox, dtype, (n, *tail)).copy_(piece)
                else:
                    self._push(self._peers[r], 0, piece, self._streams[r])
            self._sync()
        self.dist.barrier(group=self.group)  # shards have landed
        n = sizes[self.rank]
        return self._view(self.inbox, dtype, (n, *tail))

    def gather(self, local, rows_total):
        """Every rank writes its rows into its slot of the root's gather buffer; the
        root returns the concatenation (a copy), the others ``None``."""
        import torch

        sizes = self.shard_sizes(rows_total)
        if self.rank == self.root:
            if sizes[self.rank]:
                self._view(self.gather_buf, local.dtype, tuple(local.shape), self.rank * self.out_bytes).copy_(local)
        elif sizes[self.rank]:
            self._push(self._root_gather, self.rank * self.out_bytes, local.contiguous(), self._streams[self.root])
            self._sync()
        self.dist.barrier(group=self.group)  # every slot is written
        out = None
        if self.rank == self.root:
            tail = tuple(local.shape[1:])
            parts = [self._view(self.gather_buf, local.dtype, (n, *tail), r * self.out_bytes)
                     for r, n in enumerate(sizes) if n]
            out = torch.cat(parts) if parts else local[:0].clone()
        self.dist.barrier(group=self.group)  # the root has read the slots: they may be overwritten
        return out

    def close(self):
        hipshm = self._hipshm
        self.dist.barrier(group=self.group)
        for m in list(self._peers.values()) + ([self._root_gather] if self._root_gather is not None else []):
            hipshm.close_shared_memory_region(m)
        self._peers, self._root_gather = {}, None
        self.dist.barrier(group=self.group)  # every mapping is closed before its owner frees the memory
        for s in self._streams.values():
            if s is not None:
                s.close()
        self._streams = {}
        hipshm.destroy_shared_memory_region(self.inbox)
        if self.gather_buf is not None:
            hipshm.destroy_shared_memory_region(self.gather_buf)
