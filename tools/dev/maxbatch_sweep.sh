#!/bin/bash
# Headline bench against the batcher's largest merged batch.
set -u
for mb in "$@"; do
  echo -n "max_batch=$mb -> "
  TCA_RESNET_MAX_BATCH=$mb timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['p50_latency_us'], d['p99_latency_us'])" || exit 1
done
