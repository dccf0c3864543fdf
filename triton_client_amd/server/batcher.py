"""Dynamic batcher + asynchronous completion for GPU models.

Requests submitted by transport threads are merged along the batch axis (up to
``max_batch_size`` or ``max_queue_delay_us``) and executed as ONE forward on
the model's HIP stream.  Enqueue and completion are decoupled: the batcher
thread records a HIP event after enqueueing a batch and immediately starts
forming the next one while a completer thread waits for the event and wakes
the callers, so host-side request handling overlaps GPU execution.

A batch made of a single request whose tensors live in hipshm regions runs
zero-copy: the captured graph reads the client's input region and writes the
client's output region directly.
"""

from __future__ import annotations

import os
import queue
import threading
import time

import numpy as np

from ..models import ModelError
from ..utils import trace as _trace
from .core import ServerError


def _profiled(fn, tag):
    """With TCA_PROFILE_DIR set, run a thread's main function under cProfile and
    write ``<dir>/<tag>.prof`` when it returns (developer aid for finding
    per-request Python cost)."""
    import os

    out = os.environ.get("TCA_PROFILE_DIR")
    if not out:
        return fn

    def run():
        import cProfile

        prof = cProfile.Profile()
        try:
            prof.runcall(fn)
        finally:
            os.makedirs(out, exist_ok=True)
            prof.dump_stats(os.path.join(out, f"{tag}.prof"))

    return run


class _Item:
    __slots__ = ("req", "inputs", "batch", "stable", "requested", "ctx", "stats", "t_arrive", "version", "done",
                 "response", "error", "t_queued", "callback", "gap_ns", "release", "dests")

    def __init__(self, req, inputs, batch, stable, requested, ctx, stats, t_arrive, version, callback=None):
        self.req, self.inputs, self.batch, self.stable = req, inputs, batch, stable
        self.requested, self.ctx, self.stats, self.t_arrive, self.version = requested, ctx, stats, t_arrive, version
        # blocking submitters wait on the event; asynchronous ones get a callback
        self.done = threading.Event() if callback is None else None
        self.callback = callback
        self.response = None
        self.error = None
        self.t_queued = time.perf_counter_ns()
        self.gap_ns = 1 << 62  # time since the previous arrival for the same model (set by the batcher)
        self.release = None  # gives back the request's shared-memory region leases
        self.dests = None  # output name -> destination view in a device region, when the submitter knows them already

    def finish(self):
        """Hand the outcome to whoever submitted the request.  The GPU work and the copies of
        this request are complete here, so its shared-memory regions may be unmapped again."""
        if self.release is not None:
            self.release(self.req)
        if self.callback is None:
            self.done.set()
            return
        try:
            self.callback(self.response, self.error)
        except Exception:  # a broken transport callback must not stop the completer
            import traceback

            traceback.print_exc()


class DynamicBatcher:
    def __init__(self, core, lm):
        self.core = core
        self.lm = lm
        self.model = lm.model
        self.max_batch = max(1, self.model.max_batch_size)
        self.delay_s = float(getattr(self.model, "max_queue_delay_us", 0)) / 1e6
        # Burst coalescing: in a closed loop the answers of one batch go out together, so
        # the next requests come back as a burst tens of microseconds apart.  When the last
        # arrival followed its predecessor within `burst_gap_s`, or fewer requests have
        # come back than the batch answered within the last `burst_gap_s` held, an idle
        # GPU waits up to `burst_window_s` for the rest of the burst; serial traffic (one
        # request at a time, gaps of a whole round trip) never waits.
        self.burst_gap_s = float(os.environ.get("TCA_BURST_GAP_US", getattr(self.model, "burst_gap_us", 300))) / 1e6
        self.burst_window_s = float(os.environ.get("TCA_BURST_WINDOW_US", getattr(self.model, "burst_window_us",
                                                                                  100))) / 1e6
        # While every execution instance has at least one batch enqueued, a batch that is not full yet may
        # wait this long for more requests (no GPU time is lost: its instance is busy anyway).
        self.fill_window_s = float(os.environ.get("TCA_FILL_WINDOW_US", getattr(self.model, "fill_window_us", 0))) / 1e6
        # With few images outstanding (a closed loop of low concurrency: fewer than 0.6 x four full batches
        # queued or on the GPU) merged batches are capped at a quarter of them, so that every instance has one
        # batch running and one queued instead of one large batch each and nothing behind it.  Measured at
        # concurrency 4 / 8 / 16 / 32: 28.1 -> 31.7k, 38.2 -> 42.8k, 52.4 -> 57.3k, 64.4 -> 70.2k infer/s and
        # p99 1.59 -> 1.04, 2.41 -> 1.56, 2.55 -> 2.50, 5.85 -> 3.86 ms; concurrency 2 and 64 unchanged
        # (TCA_ADAPTIVE_CAP=0 turns it off).
        self._adaptive_cap = os.environ.get("TCA_ADAPTIVE_CAP", "1") != "0"
        self._images_on_gpu = 0          # launched, not completed (batcher adds, completers subtract)
        self._outstanding_peak = 0.0     # decaying maximum of images queued + on the GPU
        self._last_arrival_ns = 0
        self._last_done = (0, 1)  # (time, requests) of the batch that completed last
        # Execution instances of the model (own HIP stream + own captured graphs each):
        # batches are spread over them and run side by side on the GPU.
        self.instances = int(getattr(self.model, "instance_count", 1)) if hasattr(self.model, "stream_of") else 1
        self._busy = [0] * self.instances  # batches enqueued per instance (batcher adds, completers subtract)
        self.split_min_batch = int(os.environ.get("TCA_SPLIT_MIN_BATCH", getattr(self.model, "split_min_batch", 16)))
        self._busy_lock = threading.Lock()
        self._pins = {}  # input address of a single zero-copy request -> instance that captured its graph first
        self._launch_hist = {} if os.environ.get("TCA_BATCHER_STATS") else None
        # Models that ask for it (``pad_batches``) run a batch on the static buffers of the next size their
        # graphs were captured for at load time (rows past the batch are computed and ignored): no request,
        # whatever its batch size, makes the server capture a graph while it is serving.
        sizes = sorted(b for b in getattr(self.model, "warmup_batches", ()) if b <= self.max_batch)
        self._pad_sizes = sizes if (sizes and getattr(self.model, "pad_batches", False)
                                    and getattr(self.model, "use_graphs", False)) else None
        # per instance: [ring of [pinned uint8 buffer, event of its last H2D copy, bytes used], next entry]
        self._stagers = [None] * self.instances  # per instance: native wire stager (pinned ring + copy stream)
        self.q = queue.Queue()
        self.completions = [queue.Queue() for _ in range(self.instances)]
        self._held = None
        self._stop = False
        self._static = {}
        # at most this many batches are enqueued on the GPU at once (two per instance: one
        # running, one queued behind it): while they run, arriving requests accumulate
        # into ONE larger next batch instead of a train of small ones
        cap = int(os.environ.get("TCA_MAX_INFLIGHT_BATCHES", 0)) or int(
            getattr(self.model, "max_inflight_batches", 2 * self.instances))
        self._inflight = threading.Semaphore(cap)
        self._prewarm()
        self._threads = [
            threading.Thread(target=_profiled(self._batch_loop, "batcher"), name="tca-batcher", daemon=True)]
        for inst in range(self.instances):
            # one completer per instance: a batch that finishes first on one stream is
            # answered without waiting for an older batch on another
            self._threads.append(threading.Thread(
                target=_profiled(lambda inst=inst: self._complete_loop(inst), f"completer{inst}"),
                name=f"tca-completer-{inst}", daemon=True))
        for t in self._threads:
            t.start()

    def _prewarm(self):
        """Capture the model's HIP graphs for the merged batch sizes up front so
        no request ever waits for a capture."""
        import torch

        model = self.model
        sizes = [b for b in getattr(model, "warmup_batches", ()) if b <= self.max_batch]
        if not sizes or not getattr(model, "use_graphs", False):
            return
        spec = model.inputs[0]
        from ..utils import triton_to_np_dtype

        np_dt = triton_to_np_dtype(spec.datatype)
        t_dt = torch.from_numpy(np.zeros(1, dtype=np_dt)).dtype
        for inst in range(self.instances):
            for b in sizes:
                sbuf, obuf = self._static_buffers(inst, b, spec.name, t_dt, tuple(spec.dims), model.outputs[0].name)
                sbuf.zero_()
                model.execute({spec.name: sbuf, "__out__" + model.outputs[0].name: obuf, "__instance__": inst}, None)
            self._stream_of(inst).synchronize()

    def _stream_of(self, inst):
        return self.model.stream_of(inst) if self.instances > 1 else self.model.stream

    def _note_launch(self, inst, total):
        """Developer aid (TCA_BATCHER_STATS=<file>): how many batches each instance already had enqueued
        when a batch was launched on it, and the merged batch sizes; written when the batcher stops."""
        h = self._launch_hist
        key = (self._busy[inst] - 1, sum(self._busy) - 1, total)
        h[key] = h.get(key, 0) + 1

    def _dump_stats(self):
        path = os.environ.get("TCA_BATCHER_STATS")
        if not path or not self._launch_hist:
            return
        own, everywhere, sizes = {}, {}, {}
        for (mine, all_, total), n in self._launch_hist.items():
            own[mine] = own.get(mine, 0) + n
            everywhere[all_] = everywhere.get(all_, 0) + n
            sizes[total] = sizes.get(total, 0) + n
        with open(path, "a") as fh:
            fh.write(f"model {self.model.name}: launches {sum(own.values())}\n")
            fh.write(f"  batches already enqueued on the chosen instance: {dict(sorted(own.items()))}\n")
            fh.write(f"  batches already enqueued on all instances: {dict(sorted(everywhere.items()))}\n")
            fh.write(f"  merged batch sizes: {dict(sorted(sizes.items()))}\n")

    def _pick_instance(self, hint=None):
        """Instance for the next batch: the one with the fewest batches enqueued.
        ``hint`` (an address) keeps single zero-copy requests on the instance that already
        holds a captured graph for that client region while that instance is not the busier one."""
        with self._busy_lock:
            inst = min(range(self.instances), key=self._busy.__getitem__)
            if hint is not None:
                # the instance this address ran on first, unless another one has less to do right now (two
                # serial clients then run side by side instead of taking turns on one instance; address
                # bits would put every 2 MiB-aligned IPC mapping on instance 0)
                pinned = self._pins.get(hint)
                if pinned is None:
                    if len(self._pins) >= 4096:
                        # drop the oldest half (insertion order), not everything: the regions in use keep
                        # their instance, and with it the graphs already captured for their addresses
                        for key in list(self._pins)[:2048]:
                            del self._pins[key]
                    pinned = self._pins[hint] = inst
                if self._busy[pinned] <= self._busy[inst]:
                    inst = pinned
            self._busy[inst] += 1
        return inst

    def _instance_done(self, inst):
        with self._busy_lock:
            self._busy[inst] -= 1
        self._inflight.release()

    def stop(self):
        self._dump_stats()
        self._stop = True
        self.q.put(None)
        for c in self.completions:
            c.put(None)

    # called on transport threads
    def submit(self, req, inputs, batch, stable, requested, ctx, stats, t_arrive, version):
        item = _Item(req, inputs, max(1, batch), stable, requested, ctx, stats, t_arrive, version)
        item.release = self.core.release_request
        self._stamp(item)
        self.q.put(item)
        item.done.wait()
        if item.error is not None:
            raise item.error
        return item.response

    def submit_async(self, req, inputs, batch, stable, requested, ctx, stats, t_arrive, version, callback, dests=None):
        """Queue a request and return at once; ``callback(response, error)`` runs
        on the completer thread (or the batcher thread when the launch fails)."""
        item = _Item(req, inputs, max(1, batch), stable, requested, ctx, stats, t_arrive, version, callback)
        item.release = self.core.release_request
        item.dests = dests
        self._stamp(item)
        self.q.put(item)

    def _stamp(self, item):
        # unsynchronised on purpose: a lost update only blurs a heuristic
        now = item.t_queued
        item.gap_ns = now - self._last_arrival_ns
        self._last_arrival_ns = now

    # ------------------------------------------------------------ batching
    def _next_group(self):
        """Form the next batch.  Everything already queued joins it.  While the GPU
        has no free slot, waiting costs nothing, so arrivals keep being absorbed
        until one opens.  Once a slot is free the batch launches at once, unless the
        model asks for a queue delay or the arrivals look like a burst (see
        ``burst_gap_s``): then it waits that long, at most, for more requests.  An
        idle GPU never sits on a lone request of a serial client."""
        first = self._held or self.q.get()
        self._held = None
        if first is None:
            return None
        group, total = [first], first.batch
        t_first = time.perf_counter()
        deadline = t_first + self.delay_s
        have_slot = False
        cap = self.max_batch
        if self._adaptive_cap and first.stable:  # tensors already on the GPU: a smaller batch costs no extra staging
            seen = self._images_on_gpu + (self.q.qsize() + 1) * first.batch
            self._outstanding_peak = max(seen, self._outstanding_peak * 0.98)
            if self._outstanding_peak <= 0.6 * 4 * self.max_batch:  # clearly fewer than four full batches around
                step = max(1, first.batch)
                cap = int(-(-self._outstanding_peak / (2 * self.instances) // step) * step)
                cap = min(self.max_batch, max(step, cap))
        while total < cap:
            try:
                nxt = self.q.get_nowait()
            except queue.Empty:
                if not have_slot:
                    have_slot = self._inflight.acquire(blocking=False)
                if not have_slot:
                    # GPU full: block on the slot (short timeout so arrivals are still absorbed)
                    have_slot = self._inflight.acquire(timeout=0.0005)
                    continue
                done_ns, done_requests = self._last_done
                if (group[-1].gap_ns < self.burst_gap_s * 1e9
                        or (len(group) < done_requests
                            and time.perf_counter_ns() - done_ns < self.burst_gap_s * 1e9)):
                    # part of a burst, or the first requests coming back after a batch of
                    # several was answered: more is about to arrive
                    deadline = max(deadline, t_first + self.burst_window_s)
                filling = self.fill_window_s > 0 and min(self._busy) >= 1
                # every instance still has a batch to run: nothing idles while this one fills up; looked at
                # again every 200 us, so an instance that runs out of work gets this batch at once
                remaining = (max(deadline, t_first + self.fill_window_s) if filling else deadline) - time.perf_counter()
                if remaining <= 0:
                    break
                try:
                    nxt = self.q.get(timeout=min(remaining, 0.0002) if filling else remaining)
                except queue.Empty:
                    if filling:
                        continue
                    break
            if nxt is None:
                self._stop = True
                break
            if total + nxt.batch > cap:
                self._held = nxt
                break
            group.append(nxt)
            total += nxt.batch
        if not have_slot:
            self._inflight.acquire()
        return group, total

    def _batch_loop(self):
        import torch

        torch.cuda.set_device(self.lm.device_index or 0)
        while not self._stop:
            got = self._next_group()
            if got is None:
                break
            group, total = got
            for part, part_total in self._split_for_idle_instances(group, total):
                self._launch_guarded(part, part_total)

    def _split_for_idle_instances(self, group, total):
        """With two execution instances idle, a batch of several requests runs faster as two
        halves side by side (the halves fill each other's tails: 64 images 1.36 -> 1.27 ms)
        than as one batch on one instance.  Only while nothing else is running, so a loaded
        server keeps forming the largest batches."""
        if self.instances < 2 or len(group) < 2 or total < self.split_min_batch:
            return [(group, total)]
        with self._busy_lock:
            idle = all(b == 0 for b in self._busy)
        if not idle or not self._inflight.acquire(blocking=False):  # the second half needs its own slot
            return [(group, total)]
        first, acc = [], 0
        for it in group:
            if acc >= total / 2 and first:
                break
            first.append(it)
            acc += it.batch
        second = group[len(first):]
        if not second:
            self._inflight.release()
            return [(group, total)]
        return [(first, acc), (second, total - acc)]

    def _launch_guarded(self, group, total):
        try:
            self._launch(group, total)
        except (ServerError, ModelError) as e:
            self._inflight.release()
            err = e if isinstance(e, ServerError) else ServerError(str(e), 400)
            for it in group:
                it.error = err
                it.stats.record_fail(time.perf_counter_ns() - it.t_arrive)
                it.finish()
        except Exception as e:  # keep serving after an unexpected failure
            self._inflight.release()
            err = ServerError(f"internal error while executing '{self.model.name}': {type(e).__name__}: {e}", 500)
            for it in group:
                it.error = err
                it.finish()

    def _static_buffers(self, inst, total, in_name, in_dtype, in_tail, out_name):
        import torch

        key = (inst, total)
        bufs = self._static.get(key)
        if bufs is None:
            device = torch.device("cuda", self.lm.device_index or 0)
            bufs = (
                torch.empty((total, *in_tail), device=device, dtype=in_dtype),
                torch.empty((total, *self.model.outputs[0].dims), device=device, dtype=torch.float32),
            )
            self._static[key] = bufs
        return bufs

    def _padded(self, total):
        if self._pad_sizes is not None:
            for size in self._pad_sizes:
                if size >= total:
                    return size
        return total

    def _zero_copy_single(self, it, in_name):
        """A lone request whose tensors are in registered device regions replays a graph captured for exactly
        those addresses (no copy at all) - while the native executor does not serve the model: it takes such
        requests itself, and graphs captured per client region are captures made while serving."""
        if getattr(self.lm, "native_graphs", None) is not None:
            return False
        return in_name in it.stable and all(
            (it.dests[ro.name] if it.dests is not None else None) is not None or ro.shm is not None for ro in it.requested)

    def _launch(self, group, total):
        hint = None
        if self.instances > 1 and len(group) == 1:
            in_name = self.model.inputs[0].name
            x = group[0].inputs.get(in_name)
            if in_name in group[0].stable and hasattr(x, "data_ptr"):
                hint = x.data_ptr()
        inst = self._pick_instance(hint)
        if self._launch_hist is not None:
            self._note_launch(inst, total)
        try:
            with _trace.range(f"batch {self.model.name} n={len(group)} b={total} i={inst}"):
                return self._launch_inner(group, total, inst)
        except BaseException:
            with self._busy_lock:
                self._busy[inst] -= 1
            raise

    def _launch_inner(self, group, total, inst):
        import torch

        model, core, lm = self.model, self.core, self.lm
        out_specs = {t.name: t for t in model.outputs}
        in_name = model.inputs[0].name
        out_name = model.outputs[0].name
        t_exec = time.perf_counter_ns()
        stream = self._stream_of(inst)
        pending = []  # (item, dev_inputs or None, host copies)
        if len(group) == 1 and (self._pad_sizes is None or self._zero_copy_single(group[0], in_name)):
            it = group[0]
            host_x = it.inputs[in_name]
            staged = isinstance(host_x, np.ndarray) and host_x.dtype != np.object_
            if staged:
                # wire input: pinned staging straight into the static buffer the captured
                # graph reads (stable address), no intermediate device tensor
                dev_inputs = core._to_device(lm, {k: v for k, v in it.inputs.items() if k != in_name}, it.stable)
                x_dtype = torch.from_numpy(host_x[:0].copy()).dtype
                sbuf, _ = self._static_buffers(inst, it.batch, in_name, x_dtype, tuple(host_x.shape[1:]), out_name)
                self._stage_wire([(host_x, 0, it.batch)], sbuf, stream, inst)
                dev_inputs[in_name] = sbuf
            else:
                dev_inputs = core._to_device(lm, it.inputs, it.stable)
            dev_inputs["__instance__"] = inst
            for ro in it.requested:
                dest = (it.dests[ro.name] if it.dests is not None
                        else core._output_destination(lm, ro, out_specs[ro.name], it.batch))
                if dest is not None:
                    dev_inputs["__out__" + ro.name] = dest
            x = dev_inputs[in_name]
            if not staged and in_name not in it.stable:
                # device tensor that is not in a registered region: copy into the static
                # buffer so the captured graph sees a stable address
                sbuf, _ = self._static_buffers(inst, it.batch, in_name, x.dtype, tuple(x.shape[1:]), out_name)
                with torch.cuda.stream(stream):
                    sbuf.copy_(x, non_blocking=True)
                dev_inputs[in_name] = sbuf
            if ("__out__" + out_name) not in dev_inputs:
                _, obuf = self._static_buffers(inst, it.batch, in_name, dev_inputs[in_name].dtype,
                                               tuple(dev_inputs[in_name].shape[1:]), out_name)
                dev_inputs["__out__" + out_name] = obuf
                result = model.execute(dev_inputs, it.ctx)
                result = self._host_copies(it, result, stream, own_static=True)
                dev_inputs.pop("__out__" + out_name)
            else:
                result = model.execute(dev_inputs, it.ctx)
            pending.append((it, dev_inputs, result))
        else:
            first = group[0].inputs[in_name]
            if isinstance(first, np.ndarray):
                in_dtype = torch.from_numpy(first[:0].copy()).dtype
            else:
                in_dtype = first.dtype
            tail = tuple(first.shape[1:])
            sbuf, obuf = self._static_buffers(inst, self._padded(total), in_name, in_dtype, tail, out_name)
            from ..ops.copy import batched_copy

            # gather: device-resident inputs (hipshm) go through ONE batched copy
            # kernel, wire inputs through pinned H2D copies
            row = 0
            row_bytes = sbuf[0].numel() * sbuf.element_size()
            gather = []
            host_pairs = []
            with torch.cuda.stream(stream):
                for it in group:
                    x = it.inputs[in_name]
                    if isinstance(x, np.ndarray):
                        host_pairs.append((x, row, it.batch))
                    elif x.is_cuda and x.is_contiguous() and x.dtype == sbuf.dtype:
                        gather.append((x.data_ptr(), sbuf.data_ptr() + row * row_bytes, it.batch * row_bytes))
                    else:
                        sbuf[row: row + it.batch].copy_(x, non_blocking=True)
                    row += it.batch
            self._stage_wire(host_pairs, sbuf, stream, inst)
            batched_copy(gather, stream)
            merged = {in_name: sbuf, "__out__" + out_name: obuf, "__instance__": inst}
            out = model.execute(merged, group[0].ctx)[out_name]
            # scatter the output rows into the requests' shm destinations, again in one launch
            row = 0
            out_row_bytes = out[0].numel() * out.element_size()
            scatter = []
            host_all = None  # the whole output batch in page-locked memory: ONE D2H copy for all wire answers
            for it in group:
                piece = out[row: row + it.batch]
                dev_inputs = {}
                placed = False
                for ro in it.requested:
                    dest = (it.dests[ro.name] if it.dests is not None
                        else core._output_destination(lm, ro, out_specs[ro.name], it.batch))
                    if dest is not None:
                        if dest.dtype == out.dtype and dest.is_contiguous():
                            scatter.append((out.data_ptr() + row * out_row_bytes, dest.data_ptr(),
                                            it.batch * out_row_bytes))
                        else:
                            with torch.cuda.stream(stream):
                                dest.copy_(piece, non_blocking=True)
                        dev_inputs["__out__" + ro.name] = dest
                        placed = True
                row += it.batch
                if placed:
                    result = {out_name: dev_inputs["__out__" + out_name]}
                elif any(ro.class_count for ro in it.requested):
                    result = self._host_copies(it, {out_name: piece}, stream, own_static=True)
                else:
                    if host_all is None:
                        host_all = torch.empty((total, *out.shape[1:]), dtype=out.dtype, pin_memory=True)
                        with torch.cuda.stream(stream):
                            host_all.copy_(out[:total], non_blocking=True)  # not the padding rows
                    result = {out_name: host_all[row - it.batch: row]}
                pending.append((it, dev_inputs, result))
            batched_copy(scatter, stream)
        event = torch.cuda.Event()
        event.record(stream)
        with self._busy_lock:
            self._images_on_gpu += total
        self.completions[inst].put((event, pending, total, t_exec, time.perf_counter_ns()))

    def _stage_wire(self, items, sbuf, stream, inst):
        """Wire tensors ``[(host array, first row, rows), ...]`` -> rows of the static input
        buffer ``sbuf`` that the captured graph reads.

        Host side: one CPU memcpy per array into a small ring of page-locked buffers (a copy
        from pageable memory would be staged by the runtime in small synchronous chunks on
        this thread), then an asynchronous H2D copy.  The H2D copies run on the instance's
        COPY stream into a device-side staging buffer, and the compute stream only does a
        device-to-device copy of the finished rows (microseconds) once they have landed: the
        upload of the next batch overlaps the forward of the current one instead of queueing
        behind it on the compute stream, and the pinned ring is recycled at H2D speed.
        The whole sequence is one native call (``csrc/hip/wire_stager.cc``) that runs without
        the interpreter lock; this thread only decides which array goes to which rows."""
        if not items:
            return
        from ..ops.staging import WireStager, as_staging_source

        arrays = [as_staging_source(arr) for arr, _, _ in items]
        row_bytes = sbuf[0].numel() * sbuf.element_size()
        stager = self._stagers[inst]
        if stager is None:
            # a few ring entries of one request each: the ring then stays resident in the CPU's
            # last-level cache, which was measured to make the staging memcpy (and with it the
            # whole wire path) almost twice as fast as cycling through batch-sized buffers
            stager = self._stagers[inst] = WireStager(sbuf.device.index or 0, max(a.nbytes for a in arrays),
                                                      self.max_batch * row_bytes, entries=4)
        stager.stage(arrays, [row * row_bytes for _, row, _ in items], sbuf.data_ptr(), stream)

    def _host_copies(self, it, result, stream, own_static):
        """Enqueue D2H copies (pinned) for outputs that go back over the wire,
        so the static output buffer can be reused by the next batch."""
        import torch

        out = {}
        needs_topk = {ro.name: ro.class_count for ro in it.requested}
        with torch.cuda.stream(stream):
            for name, value in result.items():
                k = needs_topk.get(name, 0)
                if k:
                    from ..ops import classification

                    if self.lm.label_table is None and self.model.labels:
                        self.lm.label_table = classification.LabelTable(self.model.labels, value.device)
                    host, shape = classification.classification_payload_async(value, k, self.lm.label_table)
                    out[name] = ("packed", host, shape)
                else:
                    host = torch.empty(value.shape, dtype=value.dtype, pin_memory=True)
                    host.copy_(value, non_blocking=True)
                    out[name] = host
        return out

    # ---------------------------------------------------------- completion
    def _complete_loop(self, inst):
        import torch

        torch.cuda.set_device(self.lm.device_index or 0)
        drained = 0
        while True:
            entry = self.completions[inst].get()
            if entry is None:
                break
            event, pending, total, t_exec, t_enq = entry
            event.synchronize()
            self._instance_done(inst)
            with self._busy_lock:
                self._images_on_gpu -= total
            t_done = time.perf_counter_ns()
            self._last_done = (t_done, len(pending))  # closed-loop clients send this many again shortly
            out_specs = {t.name: t for t in self.model.outputs}
            for it, dev_inputs, result in pending:
                try:
                    final = {}
                    for name, value in result.items():
                        if isinstance(value, tuple) and value[0] == "packed":
                            from ..ops.classification import finish_payload
                            from .core import PackedBytes

                            final[name] = PackedBytes(*finish_payload(value[1], value[2]))
                        elif hasattr(value, "is_cuda") and not value.is_cuda:
                            final[name] = value.numpy()
                        else:
                            final[name] = value
                    requested = it.requested
                    # classification already applied above: strip the flag for _build_response
                    resp = self._respond(it, final, requested, out_specs, dev_inputs)
                    it.response = resp
                    t_out = time.perf_counter_ns()
                    it.stats.record(it.batch, t_exec - it.t_arrive, t_enq - t_exec, t_done - t_enq, t_out - t_done,
                                    executions=1 if it is pending[0][0] else 0)
                except ServerError as e:
                    it.error = e
                except Exception as e:
                    it.error = ServerError(f"internal error: {type(e).__name__}: {e}", 500)
                it.finish()
            # The HIP runtime keeps about a kilobyte per asynchronous host<->device copy until the STREAM is
            # synchronised; the event waits above do not release it.  After the answers are out, every 64th batch:
            # the wait is for the batch this loop would wait for next anyway.
            drained += 1
            if drained % 64 == 0:
                self._stream_of(inst).synchronize()

    def _respond(self, it, final, requested, out_specs, dev_inputs):
        from .core import InferResponse, OutputTensor, PackedBytes

        resp = InferResponse(model_name=it.req.model_name, model_version=it.version, id=it.req.id)
        for ro in requested:
            if ro.name not in final:
                continue
            value = final[ro.name]
            spec = out_specs[ro.name]
            key = "__out__" + ro.name
            if dev_inputs and key in dev_inputs and value is dev_inputs[key]:
                resp.outputs.append(OutputTensor(ro.name, spec.datatype, list(value.shape), None, ro.shm))
                continue
            datatype = "BYTES" if ro.class_count else spec.datatype
            if not isinstance(value, (np.ndarray, PackedBytes)):
                value = value.detach().cpu().numpy()
            if ro.shm is not None:
                self.core._write_shm(ro, datatype, value)
                resp.outputs.append(OutputTensor(ro.name, datatype, list(value.shape), None, ro.shm))
            else:
                resp.outputs.append(OutputTensor(ro.name, datatype, list(value.shape), value, None, ro.binary))
        return resp
