"""Fan-out / gather of batched tensors across GPUs over RCCL (xGMI).

Topology this is shaped for: 8 MI355X in a full mesh, 7 point-to-point xGMI
links (~153 GB/s each way) per GPU.  A ring collective only ever drives one
link per direction per GPU, so:

* **scatter** (``mode="scatter"``) – the ingest rank sends each peer ITS shard
  directly.  RCCL lowers ``scatter`` to a group of point-to-point sends, which
  go out over the 7 links concurrently: the whole batch leaves the root in
  ``(B/N * bytes) / link_bw`` instead of ``B * bytes / link_bw``.
* **broadcast** (``mode="broadcast"``) – every rank receives the whole batch
  (replicated inference, e.g. ensembles); ring bound, kept for completeness.
* **gather** – outputs are tiny (``B/N x 1000`` fp32 = 32 KB for B/N = 8) and
  therefore latency bound: one ``all_gather_into_tensor`` (single launch) beats
  N point-to-point messages.  ``gather_to_root`` is offered for when only the
  ingest rank needs the result.

Messages are exchanged as ONE contiguous tensor per call (no per-sample
collectives); callers that stream many small requests should coalesce them
first (the server's dynamic batcher does).  Works on any ``torch.distributed`` backend:
"nccl" (RCCL) on GPUs, "gloo" on CPUs for tests.
"""

from __future__ import annotations


class BatchFanout:
    """Shard the leading (batch) axis of tensors over the ranks of ``group``.

    Parameters
    ----------
    group : torch.distributed.ProcessGroup or None (default group)
    root : int
        Ingest rank (owns the full batch before ``scatter`` / after ``gather``).
    mode : "scatter" | "broadcast"
    """

    def __init__(self, group=None, root=0, mode="scatter"):
        import torch.distributed as dist

        if mode not in ("scatter", "broadcast"):
            raise ValueError("mode must be 'scatter' or 'broadcast'")
        self.dist = dist
        self.group = group
        self.root = root
        self.mode = mode
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def shard_sizes(self, batch):
        """Rows per rank for a batch of ``batch`` rows (first ranks get the remainder)."""
        base, extra = divmod(batch, self.world)
        return [base + (1 if r < extra else 0) for r in range(self.world)]

    def fan_out(self, full, shape, dtype, device):
        """Distribute ``full`` (only read on the root; shape ``shape``) and return
        this rank's shard.  ``shape`` / ``dtype`` / ``device`` describe the full
        batch so that non-root ranks can allocate their receive buffer."""
        import torch

        dist = self.dist
        batch = shape[0]
        sizes = self.shard_sizes(batch)
        if self.mode == "broadcast":
            buf = full if self.rank == self.root else torch.empty(shape, dtype=dtype, device=device)
            dist.broadcast(buf, src=self._global(self.root), group=self.group)
            start = sum(sizes[: self.rank])
            return buf[start: start + sizes[self.rank]]
        mine = torch.empty((sizes[self.rank], *shape[1:]), dtype=dtype, device=device)
        if len(set(sizes)) == 1:
            chunks = list(full.contiguous().chunk(self.world, dim=0)) if self.rank == self.root else None
            dist.scatter(mine, scatter_list=chunks, src=self._global(self.root), group=self.group)
            return mine
        # uneven shards: explicit point-to-point burst (one send per peer, all in flight together)
        if self.rank == self.root:
            reqs, start = [], 0
            for r, n in enumerate(sizes):
                piece = full[start: start + n]
                start += n
                if r == self.root:
                    mine.copy_(piece)
                elif n:
                    reqs.append(dist.isend(piece.contiguous(), dst=self._global(r), group=self.group))
            for q in reqs:
                q.wait()
        elif sizes[self.rank]:
            dist.recv(mine, src=self._global(self.root), group=self.group)
        return mine

    def gather(self, local, batch):
        """All ranks receive the concatenation of every rank's ``local`` rows
        (``batch`` rows in total)."""
        import torch

        dist = self.dist
        sizes = self.shard_sizes(batch)
        if len(set(sizes)) == 1:
            out = torch.empty((batch, *local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
            return out
        # uneven shards: pad to the largest shard so a single equal-size collective
        # still does the job, then drop the padding rows
        widest = max(sizes)
        padded = torch.zeros((widest, *local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]].copy_(local)
        out = torch.empty((widest * self.world, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, padded, group=self.group)
        return torch.cat([out[r * widest: r * widest + n] for r, n in enumerate(sizes)], dim=0)

    def gather_padded(self, padded, out):
        """The serving form of :meth:`gather`: no allocation, one equal-size collective.  ``padded`` holds
        this rank's rows padded to the widest shard, ``out`` (``world x widest`` rows, caller owned) receives
        every rank's block; the caller knows the shard sizes and drops the padding rows."""
        self.dist.all_gather_into_tensor(out, padded, group=self.group)
        return out

    def gather_to_root(self, local, batch):
        """Only the root receives the concatenation; other ranks get ``None``."""
        import torch

        dist = self.dist
        sizes = self.shard_sizes(batch)
        if self.rank == self.root:
            pieces = [torch.empty((n, *local.shape[1:]), dtype=local.dtype, device=local.device) for n in sizes]
        else:
            pieces = None
        if len(set(sizes)) == 1:
            dist.gather(local.contiguous(), gather_list=pieces, dst=self._global(self.root), group=self.group)
        else:
            if self.rank == self.root:
                reqs = []
                for r, n in enumerate(sizes):
                    if r == self.root:
                        pieces[r].copy_(local)
                    elif n:
                        reqs.append(dist.irecv(pieces[r], src=self._global(r), group=self.group))
                for q in reqs:
                    q.wait()
            elif sizes[self.rank]:
                dist.send(local.contiguous(), dst=self._global(self.root), group=self.group)
        return torch.cat(pieces, dim=0) if self.rank == self.root else None

    def _global(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)
