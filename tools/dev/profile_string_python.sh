#!/bin/bash
# cProfile of the serving process's threads (reactor callbacks, BYTES batch thread) under
# closed-loop load on simple_string with tensors in HIP shared memory.
set -u
ROOT=$(pwd)
export TCA_PROFILE_DIR=$ROOT/gpurun_out/pyprof_string
rm -rf "$TCA_PROFILE_DIR"; mkdir -p "$TCA_PROFILE_DIR"
rm -f /tmp/srv_stdin3; mkfifo /tmp/srv_stdin3
( sleep 1000 > /tmp/srv_stdin3 ) &
HOLD=$!
python3 -m triton_client_amd.server --gpu 0 --models simple_string --http-port 18200 --grpc-port 18201 \
    --exit-on-stdin-close < /tmp/srv_stdin3 > /tmp/strsrv.out 2>/tmp/strsrv.err &
SRV=$!
for i in $(seq 1 100); do
    python3 - <<'PY' && break
import sys, urllib.request
try:
    sys.exit(0 if urllib.request.urlopen("http://127.0.0.1:18200/v2/models/simple_string/ready", timeout=1).status == 200 else 1)
except Exception:
    sys.exit(1)
PY
    sleep 1
done
unset TCA_PROFILE_DIR
timeout -k 10 120 python3 -m triton_client_amd.perf -m simple_string -u 127.0.0.1:18201 -i grpc --shared-memory hip \
    --concurrency-range 16:16 --native -p 2000 --stability-windows 2 2>&1 | tail -3
kill $HOLD 2>/dev/null
wait $SRV
rm -f /tmp/srv_stdin3
python3 tools/dev/show_prof.py "$ROOT/gpurun_out/pyprof_string" 2>&1 | cut -c1-180 > "$ROOT/gpurun_out/pyprof_string.txt"
rm -rf "$ROOT/gpurun_out/pyprof_string"
