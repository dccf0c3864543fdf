"""Multi-process tests of the fan-out / gather layer on the gloo backend
(world size 2 and 3, CPU) and of client-side endpoint sharding."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, batch, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from triton_client_amd.parallel import BatchFanout

        fan = BatchFanout(root=0, mode=mode)
        full = torch.arange(batch * 6, dtype=torch.float32).reshape(batch, 2, 3) if rank == 0 else None
        shard = fan.fan_out(full, (batch, 2, 3), torch.float32, "cpu")
        sizes = fan.shard_sizes(batch)
        start = sum(sizes[:rank])
        expect = torch.arange(batch * 6, dtype=torch.float32).reshape(batch, 2, 3)[start:start + sizes[rank]]
        assert torch.equal(shard, expect), (rank, shard, expect)
        local = shard.reshape(shard.shape[0], -1).sum(dim=1, keepdim=True) + rank * 1000
        everyone = fan.gather(local, batch)
        rooted = fan.gather_to_root(local, batch)
        ref_rows = []
        for r, n in enumerate(sizes):
            s0 = sum(sizes[:r])
            rows = torch.arange(batch * 6, dtype=torch.float32).reshape(batch, 6)[s0:s0 + n].sum(dim=1, keepdim=True)
            ref_rows.append(rows + r * 1000)
        ref = torch.cat(ref_rows)
        assert torch.equal(everyone, ref)
        assert (rooted is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(rooted, ref)
        q.put((rank, "ok"))
    except Exception as e:  # report instead of hanging the peers
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,batch", [(2, "scatter", 8), (2, "broadcast", 8), (3, "scatter", 8),
                                              (2, "scatter", 5)])
def test_fanout_gather_gloo(world, mode, batch):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_sharded_client_failover():
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.parallel import ShardedClient
    from triton_client_amd.server import Server
    import tritonclient.grpc as g

    a = Server(models=DEFAULT_CPU_MODELS).start()
    b = Server(models=DEFAULT_CPU_MODELS).start()
    try:
        x = np.arange(16, dtype=np.int32).reshape(1, 16)
        i0 = g.InferInput("INPUT0", [1, 16], "INT32").set_data_from_numpy(x)
        i1 = g.InferInput("INPUT1", [1, 16], "INT32").set_data_from_numpy(x)
        with ShardedClient([a.grpc_url, b.grpc_url], protocol="grpc", probe_interval_s=0.2) as sc:
            for _ in range(10):
                np.testing.assert_array_equal(sc.infer("simple", [i0, i1]).as_numpy("OUTPUT0"), x * 2)
            assert sc.sent == [5, 5]
            b.stop()
            for _ in range(6):  # every request still succeeds, on the surviving endpoint
                np.testing.assert_array_equal(sc.infer("simple", [i0, i1], client_timeout=5).as_numpy("OUTPUT0"), x * 2)
            assert sc.live_endpoints() == [a.grpc_url]
            # a model level error is not retried elsewhere
            with pytest.raises(Exception, match="unknown model"):
                sc.infer("nope", [i0, i1])
    finally:
        a.stop()
        try:
            b.stop()
        except Exception:
            pass


def test_sharded_client_sequences_and_streams():
    """Sequence affinity (state stays on one endpoint) and one stream per endpoint
    for sequences and a decoupled model."""
    import queue

    import tritonclient.grpc as g
    from triton_client_amd.models import DEFAULT_CPU_MODELS
    from triton_client_amd.parallel import ShardedClient
    from triton_client_amd.server import Server

    servers = [Server(models=DEFAULT_CPU_MODELS).start() for _ in range(3)]
    try:
        with ShardedClient([s.grpc_url for s in servers], protocol="grpc") as sc:
            # unary sequences: the server rejects a non-START request for a sequence it has not
            # seen, so any request routed away from the sequence's endpoint fails
            def seq_input(v):
                return [g.InferInput("INPUT", [1, 1], "INT32").set_data_from_numpy(np.array([[v]], dtype=np.int32))]

            totals = {}
            for sid in (101, 202, 303, 404, 505):
                for k, v in enumerate((1, 2, 3, 4)):
                    r = sc.infer("simple_sequence", seq_input(v), sequence_id=sid, sequence_start=k == 0,
                                 sequence_end=k == 3)
                    totals.setdefault(sid, []).append(int(r.as_numpy("OUTPUT")[0, 0]))
            assert all(t == [2, 2, 3, 4] for t in totals.values()), totals  # +1 on the first request
            assert sum(1 for n in sc.sent if n) >= 2, "five sequence ids should land on more than one endpoint"

            # streams: interleaved sequences + a decoupled model over per-endpoint streams
            got = queue.Queue()
            sc.start_stream(callback=lambda result, error: got.put((result, error)))
            for k, v in enumerate((5, 6, 7)):
                for sid in ("alpha", "beta", "gamma"):
                    sc.async_stream_infer("simple_sequence", seq_input(v), sequence_id=sid, sequence_start=k == 0,
                                          sequence_end=k == 2, request_id=f"{sid}_{k}")
            finals = {}
            for _ in range(9):
                result, error = got.get(timeout=30)
                assert error is None, error
                rid = result.get_response().id
                finals[rid] = int(result.as_numpy("OUTPUT")[0, 0])
            assert all(finals[f"{sid}_0"] == 6 and finals[f"{sid}_2"] == 7 for sid in ("alpha", "beta",
                                                                                       "gamma")), finals

            n = 6
            rep = [g.InferInput("IN", [n], "INT32").set_data_from_numpy(np.arange(n, dtype=np.int32)),
                   g.InferInput("DELAY", [n], "UINT32").set_data_from_numpy(np.zeros(n, dtype=np.uint32)),
                   g.InferInput("WAIT", [1], "UINT32").set_data_from_numpy(np.zeros(1, dtype=np.uint32))]
            used = {sc.async_stream_infer("repeat_int32", rep, request_id=f"rep{j}") for j in range(3)}
            values = []
            for _ in range(3 * n):
                result, error = got.get(timeout=30)
                assert error is None, error
                values.append(int(result.as_numpy("OUT")[0]))
            assert sorted(values) == sorted(list(range(n)) * 3) and len(used) == 3
            sc.stop_stream()
    finally:
        for s in servers:
            s.stop()
