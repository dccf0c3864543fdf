#!/usr/bin/env python3
"""Fan-out / gather of a batched request over N GPUs: RCCL scatter + all_gather
against plain peer-mapped copies (one stream per xGMI link).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
      tools/fanout_bench.py --batch 256

Prints, on rank 0, the time per step and the bytes/s leaving the ingest GPU for
both transports.  One rank per GPU; needs at least two GPUs."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world < 2:
        raise SystemExit("launch with torchrun on at least two GPUs")
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from triton_client_amd.parallel import BatchFanout, DirectPeerExchange

    dev = torch.device("cuda", local)
    shape = (args.batch, 3, 224, 224)
    full = torch.randn(*shape, device=dev).half() if rank == 0 else None
    rows = -(-args.batch // world)
    fan = BatchFanout(root=0, mode="scatter")
    ex = DirectPeerExchange(rows * 3 * 224 * 224 * 2, rows * 1000 * 4, device=local, root=0)
    out_rows = torch.zeros(fan.shard_sizes(args.batch)[rank], 1000, device=dev)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        return (time.perf_counter() - t0) / args.iters

    def rccl():
        fan.fan_out(full, shape, torch.float16, dev)
        fan.gather_to_root(out_rows, args.batch)

    def direct():
        ex.scatter(full, shape, torch.float16)
        ex.gather(out_rows, args.batch)

    t_rccl, t_direct = timed(rccl), timed(direct)
    sent = (args.batch - fan.shard_sizes(args.batch)[0]) * 3 * 224 * 224 * 2  # bytes that leave the root
    if rank == 0:
        print(json.dumps({"gpus": world, "batch": args.batch, "rccl_ms": round(t_rccl * 1e3, 3),
                          "direct_ms": round(t_direct * 1e3, 3), "rccl_GBps_out": round(sent / t_rccl / 1e9, 1),
                          "direct_GBps_out": round(sent / t_direct / 1e9, 1)}))
    ex.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
