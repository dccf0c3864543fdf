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
        from triton_client_amd.parallel import BatchFanout, ShardedModelGroup

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
        # sharded model group: root drives two steps then stops the others
        group = ShardedModelGroup(lambda x: x.reshape(x.shape[0], -1).sum(dim=1, keepdim=True), (2, 3),
                                  torch.float32, "cpu", root=0, mode=mode)
        if rank == 0:
            for b in (batch, batch * 2):
                data = torch.ones(b, 2, 3)
                out = group.step(data)
                assert torch.equal(out, torch.full((b, 1), 6.0))
            group.stop()
        else:
            group.serve()
        q.put((rank, "ok"))
    except Exception as e:  # report instead of hanging the peers
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,batch", [(2, "scatter", 8), (2, "broadcast", 8), (3, "scatter", 8), (2, "scatter", 5)])
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
