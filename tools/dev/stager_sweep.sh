#!/bin/bash
# Wire / system-shm throughput of the ResNet-50 server against the number of copy threads of the stager.
set -u
for t in 1 2 4 6; do
  echo "threads=$t"
  TCA_STAGER_THREADS=$t timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 2>/dev/null > gpurun_out/stager_$t.json || exit 1
  python3 - gpurun_out/stager_$t.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], {k: v["infer_per_sec"] for k, v in d["extra"].items() if "infer_per_sec" in v})
PY
done
