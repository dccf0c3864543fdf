"""One batched inference fanned out over several GPUs, as a servable execution path.

The reference is a single-process client and has no multi-device data plane; its closest
concept is one endpoint per server with client-side rotation
(``src/java/src/main/java/triton/client/endpoint/AbstractEndpoint.java:53-67``).  Here the
ingest GPU's serving process (the *root*) owns N-1 *worker* processes, one per further GPU,
each with its own replica of the model.  A batch that arrives at the root is split along the
batch axis, the shards travel over xGMI, every GPU runs its shard and the output rows come
back into the caller's output buffer.

Steady state needs no host round trip between the processes and no process-group call:

* **mailbox** (host, shared memory): the root posts ``(sequence number, slot, rows, output
  offset)`` into the worker's ring when it submits a step, so the worker's host thread can
  enqueue its side of the step while the data is still moving;
* **flags** (fine-grained device memory on the root's GPU, IPC-mapped by the workers, see
  ``csrc/hip/flags.hip``): ``ready[worker][slot]`` is
  set by the root's copy stream behind the shard copy and waited for by the worker's stream in
  front of its forward; ``done[worker][slot]`` is set by the worker's stream behind the copy of
  its output rows and waited for by the root's stream in front of the final gather copy.
  Flags are written / polled by the GPUs themselves (one-lane kernels with a time limit, or
  command-processor wait packets); the hosts never wait for each other;
* **data** (``mode="direct"``): the shards of all peers leave the root in ONE copy-kernel launch
  that stores into the peers' inboxes (hipIpc mappings), i.e. over all xGMI links at once;
  output rows come back as one small device-to-device copy per worker into the root's gather
  buffer.  ``mode="rccl"``: grouped point-to-point sends / receives of ``torch.distributed``
  (backend "nccl" is RCCL), received straight into the caller's output rows; with
  ``gather="collective"`` the output rows come back in ONE ``all_gather_into_tensor`` of blocks
  padded to the widest shard (:meth:`parallel.fanout.BatchFanout.gather_padded`) instead of one
  receive per worker - the rows are small, so the step is bound by the number of operations.

Two slots (one per execution instance of the root's model) double-buffer the inboxes and the
gather buffer, so the scatter of one step overlaps the forward of the previous one.

With ``TCA_HIPSHM_FAKE=1`` every buffer is host memory and the lanes execute immediately
(tests on machines without a GPU).  Setup (handles, sizes, the process group for "rccl") is
exchanged once over a local socket.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import sys
import threading
import time
from multiprocessing.connection import Client, Listener

import numpy as np

LINE = 64            # bytes per flag line: every flag sits in its own cache line
RING = 16            # mailbox entries per worker
SLOTS = 2            # inbox / gather slots (execution instances of the root)
STOP_ROWS = 0xFFFFFFFF
_WORDS = LINE // 4


class ShardError(RuntimeError):
    pass


_MASK = 0xFFFFFFFF


def next_seq(seq):
    """Sequence number of the step after ``seq``.  Mailbox words, flags and the completed word are
    32 bits wide and hold ``seq & 0xFFFFFFFF``; word value 0 means "never written", so numbers whose
    low word is 0 are skipped (root and workers skip the same ones and stay on the same ring index)."""
    seq += 1
    if seq & _MASK == 0:
        seq += 1
    return seq


def _behind(seq, done_word):
    """How many steps the 32-bit progress word ``done_word`` lags ``seq`` (wrap-safe; a word that is
    ahead counts as 0)."""
    lag = (seq - done_word) & _MASK
    return 0 if lag >= 0x80000000 else lag


class ControlBlock:
    """Host shared memory of one sharded group: per worker ``RING`` mailbox lines
    ``[seq, slot, rows, out_row]`` and one line with the sequence number of the last step
    the worker has finished (published by its host thread; bounds the root's run-ahead)."""

    def __init__(self, key, workers, create):
        from ..utils import shared_memory as sysshm

        self.workers = workers
        self.nbytes = LINE * workers * (RING + 1)
        self.key = key
        if create:
            self.region = sysshm.create_shared_memory_region(key.strip("/"), key, self.nbytes)
        else:
            self.region = sysshm.attach_shared_memory_region(key.strip("/") + f"_{os.getpid()}", key, self.nbytes)
        self.words = np.frombuffer(self.region.buffer(), dtype=np.uint32)
        if create:
            self.words[:] = 0

    def mail_word(self, w, i):
        return (w * RING + i) * _WORDS

    def completed_word(self, w):
        return (self.workers * RING + w) * _WORDS

    def close(self):
        from ..utils import shared_memory as sysshm

        self.words = None
        if self.region is not None:
            sysshm.destroy_shared_memory_region(self.region)
            self.region = None


class FlagBlock:
    """The flags the GPUs set and poll: fine-grained device memory on the root's GPU, mapped by
    every worker through its IPC handle (host memory when the regions are host backed).  Per
    worker and slot one ``ready`` line, one ``done`` line, one ``error`` line."""

    READY, DONE, ERROR = 0, 1, 2

    def __init__(self, hipshm, workers, device=None, name=None, opened=None):
        self.workers = workers
        self.nbytes = LINE * workers * SLOTS * 3
        self._hipshm = hipshm
        self._owner = opened is None
        self.region = opened if opened is not None else hipshm.create_shared_memory_region(
            name, self.nbytes, device, fine_grained=True)

    def offset(self, kind, w, slot):
        return ((w * SLOTS + slot) * 3 + kind) * LINE

    def ptr(self, kind, w, slot):
        return self.region.address + self.offset(kind, w, slot)

    def snapshot(self):
        """All flag words as seen from the host (one small device-to-host copy)."""
        if self.region.host_backed:
            raw = ctypes.string_at(self.region.address, self.nbytes)
            return np.frombuffer(raw, dtype=np.uint32)[::_WORDS].copy()
        return self._hipshm._read(self.region, 0, self.nbytes).view(np.uint32)[::_WORDS].copy()

    def value(self, snapshot, kind, w, slot):
        return int(snapshot[(w * SLOTS + slot) * 3 + kind])

    def close(self):
        region, self.region = self.region, None
        if region is None:
            return
        if self._owner:
            self._hipshm.destroy_shared_memory_region(region)
        else:
            self._hipshm.close_shared_memory_region(region)


# ----------------------------------------------------------------------------- lanes
class _HostLane:
    """Immediate execution on host memory (``TCA_HIPSHM_FAKE=1``)."""

    def __init__(self, timeout_ms):
        self.timeout_s = timeout_ms / 1e3

    def copies(self, triples):
        for src, dst, n in triples:
            if n:
                ctypes.memmove(dst, src, n)

    def set_flag(self, ptr, value):
        ctypes.c_uint32.from_address(ptr).value = value & _MASK

    def wait_flag(self, ptr, value, error_ptr):
        value &= _MASK
        cell = ctypes.c_uint32.from_address(ptr)
        deadline = time.monotonic() + self.timeout_s
        while ((cell.value - value) & 0xFFFFFFFF) >= 0x80000000:
            if time.monotonic() > deadline:
                ctypes.c_uint32.from_address(error_ptr).value = 1
                return
            time.sleep(0)

    def wait_for(self, other):
        pass

    def synchronize(self):
        pass


class _GpuLane:
    """A HIP stream (``torch.cuda.Stream``) taking copies and flag operations."""

    def __init__(self, stream, flag_mode, timeout_ms):
        self.stream = stream
        self.flag_mode = flag_mode
        self.timeout_ms = int(timeout_ms)

    def copies(self, triples):
        from ..ops.copy import batched_copy

        batched_copy([t for t in triples if t[2]], self.stream)

    def set_flag(self, ptr, value):
        from ..ops._device import check, lib

        check(lib().tca_flag_set(self.stream.cuda_stream, ptr, value & _MASK, self.flag_mode), "flag set")

    def wait_flag(self, ptr, value, error_ptr):
        from ..ops._device import check, lib

        # kernel flags compare with wrap-around (signed difference); the command processor's packet is a
        # plain >=, which ShardRoot.step guards by refusing to run a "stream" group up to the wrap
        check(lib().tca_flag_wait(self.stream.cuda_stream, ptr, value & _MASK, self.flag_mode, self.timeout_ms,
                                  error_ptr), "flag wait")

    def wait_for(self, other):
        self.stream.wait_stream(other.stream)

    def synchronize(self):
        self.stream.synchronize()


def shard_sizes(rows, world):
    """Rows per rank (rank 0 = root first); the first ranks take the remainder."""
    base, extra = divmod(rows, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def _flag_mode(requested, device, fake):
    """0 = kernels, 1 = command-processor packets."""
    if fake:
        return 0
    requested = (requested or os.environ.get("TCA_SHARD_FLAGS") or "kernel").lower()
    if requested == "stream":
        from ..ops._device import check, lib

        yes = ctypes.c_int()
        check(lib().tca_can_stream_wait_value(int(device), ctypes.byref(yes)), "hipDeviceGetAttribute")
        if not yes.value:
            raise ShardError("this device does not support hipStreamWaitValue32; use flags='kernel'")
        return 1
    if requested != "kernel":
        raise ValueError("flags must be 'kernel' or 'stream'")
    return 0


# ------------------------------------------------------------------------------ root
class ShardRoot:
    """Root side of a sharded group.

    Parameters
    ----------
    forward : ``callable(x_rows, out_rows, slot)`` enqueueing the root's own shard on the
        slot's compute stream (tensors are views of the step's input / output).
    world : number of GPUs (root included)
    device : the root's GPU ordinal
    sample_bytes / out_row_bytes : bytes of one input sample / one output row
    max_rows : largest batch a step may carry
    streams : per slot, the compute stream (``torch.cuda.Stream``) ``forward`` uses; None on CPU
    mode : "direct" | "rccl"
    gather : mode "rccl" only: "p2p" (default; one receive per worker) | "collective" (one all-gather);
        ``TCA_SHARD_GATHER`` when None
    first_seq : sequence number the group counts on from (tests start just below 2^32 to cross the wrap)
    """

    def __init__(self, forward, world, device, sample_bytes, out_row_bytes, max_rows, streams=None, mode="direct",
                 flags=None, timeout_ms=5000, tag=None, first_seq=0, gather=None):
        from ..utils import hip_shared_memory as hipshm

        if mode not in ("direct", "rccl"):
            raise ValueError("mode must be 'direct' or 'rccl'")
        self.forward = forward
        self.world = int(world)
        self.workers = self.world - 1
        self.device = int(device)
        self.sample_bytes, self.out_row_bytes, self.max_rows = int(sample_bytes), int(out_row_bytes), int(max_rows)
        self.mode = mode
        self.gather_mode = gather or os.environ.get("TCA_SHARD_GATHER") or "p2p"
        if self.gather_mode not in ("p2p", "collective"):
            raise ValueError("gather must be 'p2p' or 'collective'")
        if mode != "rccl":
            self.gather_mode = "p2p"
        self._fan = None
        self._gather_bufs = {}
        self.timeout_ms = int(timeout_ms)
        self.tag = tag or f"tca_shard_{os.getpid()}_{id(self) & 0xFFFF:x}"
        self._hipshm = hipshm
        self._lock = threading.Lock()
        self._seq = int(first_seq)
        self._first_seq = int(first_seq)
        self._conns = []
        self._procs = []
        self._peer_inbox = []
        self.group = None
        self.shard_rows = -(-self.max_rows // self.world)
        probe = hipshm.create_shared_memory_region(f"{self.tag}_gather", max(1, SLOTS * self.shard_rows * self.workers
                                                                             * self.out_row_bytes), self.device)
        self.gather = probe
        self.fake = probe.host_backed
        self.flag_mode = _flag_mode(flags, self.device, self.fake)
        self.control = ControlBlock(f"/{self.tag}_ctl", max(1, self.workers), create=True)
        self.flags = FlagBlock(hipshm, max(1, self.workers), self.device, f"{self.tag}_flags")
        if self._first_seq:
            # flags as the step before would have left them (a zero flag is "ahead" of numbers near the wrap)
            words = np.zeros(self.flags.nbytes // 4, dtype=np.uint32)
            for w in range(max(1, self.workers)):
                for slot in range(SLOTS):
                    for kind in (FlagBlock.READY, FlagBlock.DONE):
                        words[self.flags.offset(kind, w, slot) // 4] = self._first_seq & _MASK
            hipshm.set_shared_memory_region(self.flags.region, [words])
        self._last_check = time.monotonic()
        if self.fake:
            self.compute = [_HostLane(self.timeout_ms) for _ in range(SLOTS)]
            self.copy = self.compute
        else:
            import torch

            if streams is None:
                streams = [torch.cuda.Stream(device=self.device) for _ in range(SLOTS)]
            self.compute = [_GpuLane(s, self.flag_mode, self.timeout_ms) for s in streams]
            self.copy = [_GpuLane(torch.cuda.Stream(device=self.device), self.flag_mode, self.timeout_ms)
                         for _ in range(SLOTS)]
        self._listener = None

    # ------------------------------------------------------------------ setup
    def listen(self):
        """Open the setup socket; returns ``(address, authkey)`` for the workers."""
        self._authkey = os.urandom(16)
        self._listener = Listener(("127.0.0.1", 0), authkey=self._authkey)
        return self._listener.address, self._authkey

    def spawn_workers(self, devices, model="resnet50", warm_rows=(), env=None, startup_timeout=600.0,
                      sample_shape=(3, 224, 224), out_shape=(1000,)):
        """Start one worker process per entry of ``devices`` (GPU ordinals of ranks 1..N-1)."""
        if len(devices) != self.workers:
            raise ValueError(f"{self.workers} workers need {self.workers} devices, got {len(devices)}")
        address, authkey = self.listen()
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        child_env = dict(os.environ)
        child_env["PYTHONPATH"] = root + os.pathsep + child_env.get("PYTHONPATH", "")
        child_env["TCA_SHARD_AUTHKEY"] = authkey.hex()
        if env:
            child_env.update(env)
        for rank, dev in enumerate(devices, start=1):
            cmd = [sys.executable, "-u", "-c", "from triton_client_amd.parallel.sharded import main; main()",
                   "--rank", str(rank), "--device",
                   str(dev), "--address", f"{address[0]}:{address[1]}", "--model", model, "--warm-rows",
                   ",".join(str(r) for r in warm_rows), "--shard-rows", str(self.shard_rows), "--sample-shape",
                   ",".join(str(v) for v in sample_shape), "--out-shape", ",".join(str(v) for v in out_shape)]
            self._procs.append(subprocess.Popen(cmd, env=child_env, cwd=root))
        self.accept_workers(startup_timeout)

    def accept_workers(self, timeout=600.0, dist_port=None):
        """Handshake with every worker: collect inbox handles, hand out the control block, the
        gather buffer and (mode "rccl") the rendezvous of the process group."""
        hipshm = self._hipshm
        deadline = time.monotonic() + timeout
        # short accept timeouts: a worker that died (bad device ordinal, import error, out of memory) is
        # reported with its exit code at once instead of after the whole start-up allowance
        self._listener._listener._socket.settimeout(0.5)
        hello = {}
        while len(hello) < self.workers:
            try:
                conn = self._listener.accept()
            except OSError:  # socket.timeout
                self._check_procs()
                if time.monotonic() > deadline:
                    raise ShardError(f"only {len(hello)} of {self.workers} shard workers connected within "
                                     f"{timeout:.0f} s") from None
                continue
            info = self._recv(conn, deadline, "hello")
            hello[info["rank"]] = (conn, info)
        if sorted(hello) != list(range(1, self.world)):
            raise ShardError(f"unexpected worker ranks {sorted(hello)}")
        if self.mode == "rccl" and dist_port is None:
            import socket

            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                dist_port = s.getsockname()[1]
        for rank in range(1, self.world):
            conn, info = hello[rank]
            self._conns.append(conn)
            # opened on THIS process's GPU: the mapping (and the lazily enabled peer access) is for the
            # device whose kernels store into the inbox, the exporter's ordinal travels inside the handle
            self._peer_inbox.append(hipshm.open_shared_memory_region(
                f"{self.tag}_inbox{rank}", info["inbox"], info["inbox_bytes"],
                info["device"] if self.fake else self.device))
            conn.send({"control_key": self.control.key, "workers": self.workers, "world": self.world,
                       "flags": self.flags.region._raw, "flags_bytes": self.flags.nbytes,
                       "gather": self.gather._raw, "gather_bytes": self.gather.byte_size, "gather_device": self.device,
                       "shard_rows": self.shard_rows, "sample_bytes": self.sample_bytes,
                       "out_row_bytes": self.out_row_bytes, "mode": self.mode, "flag_mode": self.flag_mode,
                       "timeout_ms": self.timeout_ms, "dist_port": dist_port, "first_seq": self._first_seq,
                       "gather_mode": self.gather_mode})
        if self.mode == "rccl":
            import torch
            import torch.distributed as dist

            backend = "gloo" if self.fake else "nccl"
            kwargs = {} if self.fake else {"device_id": torch.device("cuda", self.device)}
            if not dist.is_initialized():
                dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{dist_port}", rank=0,
                                        world_size=self.world, **kwargs)
            self.group = dist
        for conn in self._conns:
            ack = self._recv(conn, deadline, "ready")
            if ack != "ready":
                raise ShardError(f"worker failed during setup: {ack}")
        self._listener.close()
        self._listener = None

    def _check_procs(self):
        for rank, p in enumerate(self._procs, start=1):
            if p.poll() is not None:
                raise ShardError(f"shard worker {rank} exited with code {p.returncode}")

    def _recv(self, conn, deadline, what):
        """One setup message from a worker; gives up when a worker process has exited or at the deadline."""
        while not conn.poll(0.5):
            self._check_procs()
            if time.monotonic() > deadline:
                raise ShardError(f"no '{what}' message from a shard worker before the start-up deadline")
        try:
            return conn.recv()
        except (EOFError, OSError) as e:
            self._check_procs()
            raise ShardError(f"a shard worker closed its setup connection before '{what}'") from e

    # ------------------------------------------------------------------- step
    def _post(self, w, seq, slot, rows, out_row, total=0):
        """Mailbox entry for worker ``w`` (0-based): payload first, sequence number last."""
        ctl = self.control
        # never run more than a ring ahead of the slowest worker (the batcher's in-flight cap
        # keeps it far below that; this is the backstop)
        if seq > self._first_seq + RING:
            deadline = time.monotonic() + self.timeout_ms / 1e3
            done_word = ctl.completed_word(w)
            while _behind(seq, int(ctl.words[done_word])) >= RING:
                if time.monotonic() > deadline:
                    raise ShardError(f"worker {w + 1} is more than {RING} steps behind")
                self._check_procs()
                time.sleep(0.0002)
        base = ctl.mail_word(w, seq % RING)
        ctl.words[base + 1] = slot
        ctl.words[base + 2] = rows
        ctl.words[base + 3] = out_row
        ctl.words[base + 4] = total  # rows of the whole step: the widest shard sizes the collective gather
        ctl.words[base] = seq & _MASK

    def check(self, force=True):
        """Raise if a wait on either side gave up (a process died or hangs).  The flags live in
        device memory: steps look at them a few times per second (``force=False``)."""
        self._check_procs()
        now = time.monotonic()
        if not force and now - self._last_check < 0.25:
            return
        self._last_check = now
        snap = self.flags.snapshot()
        for w in range(self.workers):
            for s in range(SLOTS):
                if self.flags.value(snap, FlagBlock.ERROR, w, s):
                    raise ShardError(f"sharded step timed out waiting for worker {w + 1} (slot {s})")

    def step(self, x, out, slot=0):
        """Enqueue one sharded inference: ``x`` ``[rows, ...]`` and ``out`` ``[rows, ...]`` on the
        root's GPU (CPU tensors in fake mode).  Everything is stream ordered on the slot's
        compute stream; the caller synchronises that stream (or an event on it)."""
        rows = int(x.shape[0])
        if rows > self.max_rows:
            raise ShardError(f"batch {rows} exceeds the group's capacity {self.max_rows}")
        self.check(force=False)
        sizes = shard_sizes(rows, self.world)
        n0 = sizes[0]
        x_ptr, out_ptr = x.data_ptr(), out.data_ptr()
        compute, copy = self.compute[slot], self.copy[slot]
        flags = self.flags
        with self._lock:  # sequence numbers are handed out in submission order
            if self.flag_mode == 1 and self._seq >= _MASK - RING:
                raise ShardError("a group with command-processor flags must be rebuilt before 2^32 steps")
            seq = self._seq = next_seq(self._seq)
            start = n0
            plan = []
            for w in range(self.workers):
                n = sizes[w + 1]
                self._post(w, seq, slot, n, start - n0, rows)
                plan.append((w, start, n))
                start += n
            if self.mode == "direct":
                copy.wait_for(compute)  # the producer of x (e.g. the batcher's gather) ran on the compute stream
                inbox_off = slot * self.shard_rows * self.sample_bytes
                copy.copies([(x_ptr + s * self.sample_bytes, self._peer_inbox[w].address + inbox_off,
                              n * self.sample_bytes) for w, s, n in plan])
                for w, _, _ in plan:
                    copy.set_flag(flags.ptr(FlagBlock.READY, w, slot), seq)
            elif self.gather_mode == "collective":
                self._rccl_collective(x, out, plan, compute, slot, sizes)
                return out
            else:
                sends = self._rccl_group(x, out, plan, compute, send=True)
                if n0:
                    self.forward(x[:n0], out[:n0], slot)
                self._rccl_wait(sends, compute)  # x may be overwritten by the next step on this stream
                self._rccl_wait(self._rccl_group(x, out, plan, compute, send=False), compute)
                return out
        if n0:
            self.forward(x[:n0], out[:n0], slot)
        if self.mode == "direct":
            for w, _, _ in plan:
                compute.wait_flag(flags.ptr(FlagBlock.DONE, w, slot), seq, flags.ptr(FlagBlock.ERROR, w, slot))
            if rows > n0:
                gather_off = slot * self.shard_rows * self.workers * self.out_row_bytes
                compute.copies([(self.gather.address + gather_off, out_ptr + n0 * self.out_row_bytes,
                                 (rows - n0) * self.out_row_bytes)])
        return out

    def peer_bandwidth(self, src_ptr, nbytes, iters=20):
        """Mode "direct": bytes per second of the copy kernel storing ``nbytes`` from ``src_ptr`` into each
        worker's inbox, one peer at a time and then all peers in one launch (what a step does).  Returns
        ``{"per_peer": [...], "all_peers": total bytes per second}``."""
        lane = self.copy[0]
        nbytes = min(int(nbytes), *(m.byte_size for m in self._peer_inbox))

        def timed(triples):
            lane.copies(triples)
            lane.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                lane.copies(triples)
            lane.synchronize()
            return sum(t[2] for t in triples) * iters / max(time.perf_counter() - t0, 1e-9)

        each = [timed([(src_ptr, m.address, nbytes)]) for m in self._peer_inbox]
        together = timed([(src_ptr, m.address, nbytes) for m in self._peer_inbox])
        return {"bytes": nbytes, "per_peer": each, "all_peers": together}

    def _rccl_group(self, x, out, plan, compute, send):
        """One group of point-to-point operations: every shard leaves in one group (all links
        at once); the output rows are received straight into ``out``.  Issued under the lock,
        in sequence order: every rank must see the same order of operations."""
        import torch

        dist = self.group
        ops = []
        for w, s, n in plan:
            if n:
                ops.append(dist.P2POp(dist.isend, x[s: s + n], w + 1) if send
                           else dist.P2POp(dist.irecv, out[s: s + n], w + 1))
        if not ops:
            return []
        if self.fake:
            return dist.batch_isend_irecv(ops)
        with torch.cuda.stream(compute.stream):
            return dist.batch_isend_irecv(ops)

    def _rccl_collective(self, x, out, plan, compute, slot, sizes):
        """Shards leave in one group of sends; the root's rows are computed into a block padded to the
        widest shard; one all-gather brings every rank's block; one batched copy drops the padding."""
        import contextlib

        import torch

        from .fanout import BatchFanout

        if self._fan is None:
            self._fan = BatchFanout(group=None, root=0)
        widest, n0 = sizes[0], sizes[0]
        key = (slot, out.dtype, tuple(out.shape[1:]))
        bufs = self._gather_bufs.get(key)
        if bufs is None:
            pad = torch.zeros((self.shard_rows, *out.shape[1:]), dtype=out.dtype, device=out.device)
            full = torch.zeros((self.world * self.shard_rows, *out.shape[1:]), dtype=out.dtype, device=out.device)
            bufs = self._gather_bufs[key] = (pad, full)
        pad, full = bufs
        sends = self._rccl_group(x, out, plan, compute, send=True)
        self.forward(x[:n0], pad[:n0], slot)
        self._rccl_wait(sends, compute)
        with contextlib.nullcontext() if self.fake else torch.cuda.stream(compute.stream):
            self._fan.gather_padded(pad[:widest], full[: self.world * widest])
        row = self.out_row_bytes
        triples, start = [], 0
        for r, n in enumerate(sizes):
            if n:
                triples.append((full.data_ptr() + r * widest * row, out.data_ptr() + start * row, n * row))
            start += n
        compute.copies(triples)

    def _rccl_wait(self, works, compute):
        if self.fake:
            for work in works:
                work.wait()
            return
        import torch

        with torch.cuda.stream(compute.stream):
            for work in works:
                work.wait()  # NCCL backend: the compute stream waits, not the host

    # -------------------------------------------------------------------- end
    def close(self):
        try:
            with self._lock:
                self._seq = next_seq(self._seq)
                for w in range(self.workers):
                    self._post(w, self._seq, 0, STOP_ROWS, 0)
        except Exception:
            pass
        for p in self._procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for conn in self._conns:
            try:
                conn.close()
            except Exception:
                pass
        for lane in self.compute + self.copy:
            lane.synchronize()
        for m in self._peer_inbox:
            try:
                self._hipshm.close_shared_memory_region(m)
            except Exception:
                pass
        self._peer_inbox = []
        if self.group is not None:
            try:
                self.group.destroy_process_group()
            except Exception:
                pass
            self.group = None
        self.control.close()
        self.flags.close()
        self._hipshm.destroy_shared_memory_region(self.gather)


# ---------------------------------------------------------------------------- worker
class ShardWorker:
    """Worker side.  ``make_forward(worker)`` returns ``(forward, streams)``:
    ``forward(x_rows, out_rows, slot)`` enqueues the replica's forward on ``streams[slot]``
    (``x_rows`` is a view of the inbox slot, ``out_rows`` a local buffer); ``streams`` is None
    on host memory."""

    def __init__(self, rank, device, address, authkey, make_forward, out_dtype="FP32", in_dtype="FP16",
                 sample_shape=(3, 224, 224), out_shape=(1000,)):
        from ..utils import hip_shared_memory as hipshm

        self._hipshm = hipshm
        self.rank, self.device = int(rank), int(device)
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.sample_shape, self.out_shape = tuple(sample_shape), tuple(out_shape)
        self.conn = Client(address, authkey=authkey)
        self._make_forward = make_forward
        self.group = None

    def setup(self, inbox_rows_hint, sample_bytes):
        """Allocate the inbox, exchange handles, attach to the control block."""
        hipshm = self._hipshm
        inbox_bytes = SLOTS * inbox_rows_hint * sample_bytes
        self.inbox = hipshm.create_shared_memory_region(f"tca_shard_inbox_{os.getpid()}", inbox_bytes, self.device)
        self.fake = self.inbox.host_backed
        self.conn.send({"rank": self.rank, "device": self.device, "inbox": self.inbox._raw, "inbox_bytes": inbox_bytes})
        cfg = self.cfg = self.conn.recv()
        if cfg["shard_rows"] > inbox_rows_hint or cfg["sample_bytes"] != sample_bytes:
            raise ShardError("the root's shard geometry does not fit this worker's inbox")
        self.shard_rows = cfg["shard_rows"]  # slot stride inside the inbox: the root's, it does the addressing
        self.sample_bytes, self.out_row_bytes = cfg["sample_bytes"], cfg["out_row_bytes"]
        self.mode = cfg["mode"]
        self.control = ControlBlock(cfg["control_key"], cfg["workers"], create=False)
        # the root's flag block and gather buffer, mapped for THIS replica's GPU (its kernels poll and store)
        own = cfg["gather_device"] if self.fake else self.device
        self.flags = FlagBlock(hipshm, cfg["workers"], opened=hipshm.open_shared_memory_region(
            f"tca_shard_flags_{os.getpid()}", cfg["flags"], cfg["flags_bytes"], own))
        self.gather = hipshm.open_shared_memory_region(f"tca_shard_gather_{os.getpid()}", cfg["gather"],
                                                       cfg["gather_bytes"], own)
        if not self.fake:
            import torch

            torch.cuda.set_device(self.device)
        self.forward, self.streams = self._make_forward(self)
        if self.fake:
            self.lanes = [_HostLane(cfg["timeout_ms"]) for _ in range(SLOTS)]
        else:
            self.lanes = [_GpuLane(self.streams[s], cfg["flag_mode"], cfg["timeout_ms"]) for s in range(SLOTS)]
        if self.mode == "rccl":
            import torch
            import torch.distributed as dist

            backend = "gloo" if self.fake else "nccl"
            kwargs = {} if self.fake else {"device_id": torch.device("cuda", self.device)}
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{cfg['dist_port']}", rank=self.rank,
                                    world_size=cfg["world"], **kwargs)
            self.group = dist
        self._views = {}
        self._inflight = []

    def _tensors(self, slot, rows):
        """(input view of the inbox slot, local output buffer) for a shard of ``rows`` rows."""
        key = (slot, rows)
        got = self._views.get(key)
        if got is None:
            import torch

            off = slot * self.shard_rows * self.sample_bytes
            x = self._hipshm.mapping_as_torch(self.inbox, self.in_dtype, [rows, *self.sample_shape], off)
            from ..utils import triton_to_np_dtype

            dt = torch.from_numpy(np.zeros(1, dtype=triton_to_np_dtype(self.out_dtype))).dtype
            out = torch.empty((rows, *self.out_shape), dtype=dt, device=x.device)
            got = self._views[key] = (x, out)
        return got

    def ready(self):
        self.conn.send("ready")

    def fail(self, text):
        try:
            self.conn.send(text)
        except Exception:
            pass

    def serve(self):
        """Process mailbox entries in sequence order until the stop entry arrives."""
        ctl = self.control
        w = self.rank - 1
        seq = next_seq(int(self.cfg.get("first_seq", 0)))
        idle_since = time.monotonic()
        while True:
            base = ctl.mail_word(w, seq % RING)
            self._publish_completed(w)
            while int(ctl.words[base]) != seq & _MASK:
                # spin briefly (a loaded group posts every few hundred microseconds), then back off
                if time.monotonic() - idle_since > 0.005:
                    time.sleep(0.0002)
                    self._publish_completed(w)
                    if self.conn.poll(0):  # the root went away
                        try:
                            self.conn.recv()
                        except (EOFError, OSError):
                            return
            slot, rows, out_row = int(ctl.words[base + 1]), int(ctl.words[base + 2]), int(ctl.words[base + 3])
            if rows == STOP_ROWS:
                return
            self._total = int(ctl.words[base + 4])
            self._step(w, seq, slot, rows, out_row)
            self._track(seq, slot)
            seq = next_seq(seq)
            idle_since = time.monotonic()

    def _step(self, w, seq, slot, rows, out_row):
        flags, lane = self.flags, self.lanes[slot]
        if self.mode == "rccl":
            self._step_rccl(slot, rows)
            return
        lane.wait_flag(flags.ptr(FlagBlock.READY, w, slot), seq, flags.ptr(FlagBlock.ERROR, w, slot))
        if rows:
            x, out = self._tensors(slot, rows)
            self.forward(x, out, slot)
            dst = (self.gather.address + slot * self.cfg["shard_rows"] * self.cfg["workers"] * self.out_row_bytes
                   + out_row * self.out_row_bytes)
            lane.copies([(out.data_ptr(), dst, rows * self.out_row_bytes)])
        lane.set_flag(flags.ptr(FlagBlock.DONE, w, slot), seq)

    def _track(self, seq, slot):
        """Remember the step's completion event; keep at most ``RING - 2`` steps enqueued."""
        if self.fake:
            self._inflight.append((seq, None))
            return
        import torch

        ev = torch.cuda.Event()
        ev.record(self.streams[slot])
        self._inflight.append((seq, ev))
        if len(self._inflight) >= RING - 2:
            self._inflight[0][1].synchronize()

    def _publish_completed(self, w):
        """Host-visible progress: sequence number of the newest step all earlier steps of which
        have finished on the GPU."""
        done = None
        while self._inflight and (self._inflight[0][1] is None or self._inflight[0][1].query()):
            done = self._inflight.pop(0)[0]
        if done is not None:
            self.control.words[self.control.completed_word(w)] = done & _MASK

    def _step_rccl(self, slot, rows):
        if self.cfg.get("gather_mode") == "collective":
            return self._step_rccl_collective(slot, rows)
        if not rows:
            return
        x, out = self._tensors(slot, rows)
        dist = self.group
        if self.fake:
            dist.recv(x, src=0)
            self.forward(x, out, slot)
            dist.send(out, dst=0)
            return
        import torch

        # Grouped calls like the root's: on the NCCL backend a grouped point-to-point operation runs on
        # the group's communicator while a plain send / recv creates one per pair of ranks - the two
        # forms never meet each other.  wait() makes this stream wait, not the host.
        with torch.cuda.stream(self.streams[slot]):
            for work in dist.batch_isend_irecv([dist.P2POp(dist.irecv, x, 0)]):
                work.wait()
            self.forward(x, out, slot)
            for work in dist.batch_isend_irecv([dist.P2POp(dist.isend, out, 0)]):
                work.wait()

    def _step_rccl_collective(self, slot, rows):
        """Every rank joins the all-gather of every step, rows or not (a step smaller than the group)."""
        import contextlib

        import torch

        from ..utils import triton_to_np_dtype
        from .fanout import BatchFanout

        dist = self.group
        world = self.cfg["world"]
        widest = -(-self._total // world)
        bufs = self._views.get(("gather", slot))
        if bufs is None:
            dt = torch.from_numpy(np.zeros(1, dtype=triton_to_np_dtype(self.out_dtype))).dtype
            device = "cpu" if self.fake else f"cuda:{self.device}"
            pad = torch.zeros((self.shard_rows, *self.out_shape), dtype=dt, device=device)
            full = torch.zeros((world * self.shard_rows, *self.out_shape), dtype=dt, device=device)
            bufs = self._views[("gather", slot)] = (pad, full, BatchFanout(group=None, root=0))
        pad, full, fan = bufs
        with contextlib.nullcontext() if self.fake else torch.cuda.stream(self.streams[slot]):
            if rows:
                x, _ = self._tensors(slot, rows)
                if self.fake:
                    dist.recv(x, src=0)
                else:
                    for work in dist.batch_isend_irecv([dist.P2POp(dist.irecv, x, 0)]):
                        work.wait()
                self.forward(x, pad[:rows], slot)
            fan.gather_padded(pad[:widest], full[: world * widest])

    def close(self):
        for lane in getattr(self, "lanes", ()):
            lane.synchronize()
        if self.group is not None:
            try:
                self.group.destroy_process_group()
            except Exception:
                pass
        self._views = {}
        self.control.close()
        self.flags.close()
        self._hipshm.close_shared_memory_region(self.gather)
        self._hipshm.destroy_shared_memory_region(self.inbox)
        self.conn.close()


def _resnet_forward_factory(model_name):
    """Worker replica of a zoo model: graphs captured per (inbox slot, rows)."""

    def make(worker):
        from ..models import MODEL_ZOO

        model = MODEL_ZOO[model_name]()
        model.instance_count = SLOTS
        model.load(worker.device)

        def forward(x, out, slot):
            model.run(x, out, slot, stable=True)

        worker._model = model
        return forward, [model.stream_of(s) for s in range(SLOTS)]

    return make


def _rowsum_forward_factory(worker):
    """Stand-in replica for tests: ``out[i, :] = sum(x[i]) + 1000 * rank``."""

    streams = None
    if not worker.fake:
        import torch

        streams = [torch.cuda.Stream(device=worker.device) for _ in range(SLOTS)]

    def forward(x, out, slot):
        if streams is None:
            out.copy_((x.float().reshape(x.shape[0], -1).sum(dim=1, keepdim=True) + 1000 * worker.rank).expand_as(out))
            return
        import torch

        with torch.cuda.stream(streams[slot]):
            out.copy_((x.float().reshape(x.shape[0], -1).sum(dim=1, keepdim=True) + 1000 * worker.rank).expand_as(out))

    return forward, streams


def _noop_forward_factory(worker):
    """Replica that computes nothing (``out`` keeps what it held): transport measurements."""
    streams = None
    if not worker.fake:
        import torch

        streams = [torch.cuda.Stream(device=worker.device) for _ in range(SLOTS)]
    return (lambda x, out, slot: None), streams


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="shard worker: one replica of a model on one GPU")
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--device", type=int, required=True)
    ap.add_argument("--address", required=True)
    ap.add_argument("--model", default="resnet50",
                    help="zoo model, or the stand-ins 'test_rowsum' (tests) / 'test_noop' (transport benchmark)")
    ap.add_argument("--warm-rows", default="")
    ap.add_argument("--shard-rows", type=int, default=int(os.environ.get("TCA_SHARD_ROWS", "128")))
    ap.add_argument("--sample-shape", default="3,224,224")
    ap.add_argument("--out-shape", default="1000")
    args = ap.parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    host, port = args.address.rsplit(":", 1)
    authkey = bytes.fromhex(os.environ["TCA_SHARD_AUTHKEY"])
    sample_shape = tuple(int(v) for v in args.sample_shape.split(","))
    out_shape = tuple(int(v) for v in args.out_shape.split(","))
    factory = {"test_rowsum": _rowsum_forward_factory, "test_noop": _noop_forward_factory}.get(args.model)
    if factory is None:
        factory = _resnet_forward_factory(args.model)
    worker = ShardWorker(args.rank, args.device, (host, int(port)), authkey, factory, sample_shape=sample_shape,
                         out_shape=out_shape)
    try:
        worker.setup(args.shard_rows, int(np.prod(sample_shape)) * 2)
        for rows in [int(r) for r in args.warm_rows.split(",") if r]:
            for slot in range(SLOTS):
                x, out = worker._tensors(slot, rows)
                worker.forward(x, out, slot)
        for lane in worker.lanes:
            lane.synchronize()
        worker.ready()
    except Exception as e:
        import traceback

        worker.fail(f"{type(e).__name__}: {e}\n{traceback.format_exc()}")
        raise
    try:
        worker.serve()
    finally:
        worker.close()


if __name__ == "__main__":
    main()
