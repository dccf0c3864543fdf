#!/bin/bash
# Headline bench against the serving knobs (execution instances, batches in flight, reactors).
set -u
run() {
  echo -n "$* -> "
  env "$@" timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['p50_latency_us'], d['p99_latency_us'])" || exit 1
}
run TCA_RESNET_INSTANCES=2 TCA_MAX_INFLIGHT_BATCHES=4
run TCA_RESNET_INSTANCES=2 TCA_MAX_INFLIGHT_BATCHES=6
run TCA_RESNET_INSTANCES=3 TCA_MAX_INFLIGHT_BATCHES=6
run TCA_RESNET_INSTANCES=2 TCA_MAX_INFLIGHT_BATCHES=4 TCA_GRPC_REACTORS=4
run TCA_RESNET_INSTANCES=2 TCA_MAX_INFLIGHT_BATCHES=4 TCA_GRPC_REACTORS=1
run TCA_RESNET_INSTANCES=2 TCA_MAX_INFLIGHT_BATCHES=3
