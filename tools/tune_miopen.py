#!/usr/bin/env python3
"""Regenerate the MIOpen find-db / perf-db shipped in
``triton_client_amd/models/miopen_db`` (run on an MI355X):

    python tools/tune_miopen.py --out gpurun_out/miopen_db

Runs the ResNet-50 instance once per batch size the dynamic batcher can form
with MIOpen's exhaustive search enabled; MIOpen records the winners in the
user database directory given by ``--out``."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--models", default="resnet50")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = out
    os.environ["TCA_MIOPEN_TUNE"] = "1"
    import torch

    from triton_client_amd.models import MODEL_ZOO

    for name in args.models.split(","):
        model = MODEL_ZOO[name]()
        model.load(0)
        for b in model.warmup_batches:
            t0 = time.time()
            x = torch.zeros(b, 3, 224, 224, device="cuda", dtype=torch.float16)
            model.run(x)
            model.stream.synchronize()
            print(f"{name} batch {b}: tuned in {time.time() - t0:.1f}s", flush=True)
        model.unload()
    print("databases:", os.listdir(out))


if __name__ == "__main__":
    main()
