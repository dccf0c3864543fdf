"""Driver of the sharded-group tests (run as a child process so that the host-backed mode of
the device shared-memory library can be chosen by environment before it is loaded).

Root: forward = row sums; workers: ``--model test_rowsum`` (row sums + 1000 * rank).  Prints
``SHARD_OK <json>`` when every step produced the expected rows."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _CountingConn:
    def __init__(self, conn, counter):
        self._conn, self._counter = conn, counter

    def __getattr__(self, name):
        attr = getattr(self._conn, name)
        if name in ("send", "recv", "send_bytes", "recv_bytes"):
            def counted(*a, **k):
                self._counter[0] += 1
                return attr(*a, **k)
            return counted
        return attr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=3)
    ap.add_argument("--mode", default="direct")
    ap.add_argument("--device", type=int, default=None, help="GPU ordinal shared by every rank (omit: host memory)")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--flags", default=None)
    ap.add_argument("--gather", default=None, help="mode rccl: p2p | collective")
    ap.add_argument("--first-seq", type=int, default=0, help="start counting steps here (just below 2^32: the wrap)")
    ap.add_argument("--kill-worker-after", type=int, default=None,
                    help="kill worker 1 after this many steps; the next steps must raise ShardError")
    args = ap.parse_args()
    import torch

    from triton_client_amd.parallel.sharded import SLOTS, ShardRoot

    fake = args.device is None
    dev = "cpu" if fake else f"cuda:{args.device}"
    sample_shape, out_cols = (2, 3), 4
    max_rows = 32
    streams = None if fake else [torch.cuda.Stream(device=dev) for _ in range(SLOTS)]

    def forward(x, out, slot):
        val = x.float().reshape(x.shape[0], -1).sum(dim=1, keepdim=True).expand_as(out)
        if streams is None:
            out.copy_(val)
        else:
            with torch.cuda.stream(streams[slot]):
                out.copy_(x.float().reshape(x.shape[0], -1).sum(dim=1, keepdim=True).expand_as(out))

    root = ShardRoot(forward, args.world, args.device or 0, 6 * 2, out_cols * 4, max_rows, streams=streams,
                     mode=args.mode, flags=args.flags, timeout_ms=4000 if args.kill_worker_after is not None else 20000,
                     first_seq=args.first_seq, gather=args.gather)
    root.spawn_workers([args.device or 0] * (args.world - 1), model="test_rowsum", sample_shape=sample_shape,
                       out_shape=(out_cols,), startup_timeout=300.0)
    # steady state: no control traffic between the processes, no process-group call (direct mode)
    conn_calls = [0]
    root._conns = [_CountingConn(c, conn_calls) for c in root._conns]
    dist_calls = [0]
    if args.mode == "direct":
        assert root.group is None
        import torch.distributed as dist

        for name in ("barrier", "broadcast", "all_gather", "all_gather_into_tensor", "all_reduce", "send", "recv",
                     "isend", "irecv", "batch_isend_irecv", "scatter", "gather", "broadcast_object_list"):
            orig = getattr(dist, name)

            def counted(*a, _orig=orig, **k):
                dist_calls[0] += 1
                return _orig(*a, **k)
            setattr(dist, name, counted)
    rows_seen = []
    pending = []
    if args.kill_worker_after is not None:
        return killed_worker(args, root, dev, sample_shape, out_cols)
    for step in range(args.steps):
        rows = [max_rows, 8, 5, 1, 2, 17][step % 6]
        slot = step % SLOTS
        ref = (torch.arange(rows * 6, dtype=torch.float32).reshape(rows, *sample_shape) % 13 + step).half()
        x = ref.to(dev)
        out = torch.zeros(rows, out_cols, dtype=torch.float32, device=dev)
        if streams is not None:
            streams[slot].wait_stream(torch.cuda.current_stream())
        root.step(x, out, slot)
        pending.append((step, rows, slot, ref, x, out))
        rows_seen.append(rows)
    for step, rows, slot, ref, x, out in pending:
        if streams is not None:
            streams[slot].synchronize()
        sizes = [rows // args.world + (1 if r < rows % args.world else 0) for r in range(args.world)]
        expect, start = [], 0
        for r, n in enumerate(sizes):
            expect.append(ref[start:start + n].float().reshape(n, 6).sum(dim=1, keepdim=True).repeat(1, out_cols)
                          + 1000 * r)
            start += n
        expect = torch.cat(expect)
        got = out.cpu()
        if not torch.equal(got, expect):
            print("MISMATCH step", step, "rows", rows, got[:, 0].tolist(), expect[:, 0].tolist())
            root.close()
            sys.exit(1)
    root.check()
    root.close()
    assert conn_calls[0] == 0, f"{conn_calls[0]} control messages in steady state"
    assert dist_calls[0] == 0, f"{dist_calls[0]} process-group calls in steady state"
    print("SHARD_OK", json.dumps({"steps": args.steps, "rows": rows_seen, "mode": args.mode,
                                  "control_messages": conn_calls[0], "process_group_calls": dist_calls[0]}))


def killed_worker(args, root, dev, sample_shape, out_cols):
    """A worker process dies in the middle of serving: within the group's time limit a step (or the
    check behind it) raises ShardError naming the worker; nothing hangs, close() still returns."""
    import time

    import torch

    from triton_client_amd.parallel.sharded import ShardError

    def one(step):
        x = (torch.arange(8 * 6, dtype=torch.float32).reshape(8, *sample_shape) % 13 + step).half().to(dev)
        out = torch.zeros(8, out_cols, dtype=torch.float32, device=dev)
        root.step(x, out, step % 2)
        root.check()
        return out

    for step in range(args.kill_worker_after):
        one(step)
    root._procs[0].kill()
    t0 = time.monotonic()
    error = None
    try:
        for step in range(64):  # far more than the ring holds
            one(args.kill_worker_after + step)
            time.sleep(0.05)
    except ShardError as e:
        error = str(e)
    waited = time.monotonic() - t0
    t1 = time.monotonic()
    root.close()
    print("SHARD_OK", json.dumps({"error": error, "seconds_to_error": round(waited, 2),
                                  "seconds_to_close": round(time.monotonic() - t1, 2)}))


if __name__ == "__main__":
    main()
