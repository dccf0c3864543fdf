#!/bin/bash
# Per-kernel time inside a serving process under closed-loop gRPC load on `simple_string` with
# BYTES tensors in HIP shared memory: shows the device BYTES path (index / add-sub / pack kernels).
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
rm -f /tmp/srv_stdin2; mkfifo /tmp/srv_stdin2
( sleep 1000 > /tmp/srv_stdin2 ) &
HOLD=$!
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/strsrv" -o s -- \
    python3 -m triton_client_amd.server --gpu 0 --models simple_string --http-port 18100 --grpc-port 18101 \
    --exit-on-stdin-close < /tmp/srv_stdin2 > "$OUT/strsrv.out" 2>"$OUT/strsrv.err" &
SRV=$!
for i in $(seq 1 100); do
    python3 - <<'PY' && break
import sys, urllib.request
try:
    sys.exit(0 if urllib.request.urlopen("http://127.0.0.1:18100/v2/models/simple_string/ready", timeout=1).status == 200 else 1)
except Exception:
    sys.exit(1)
PY
    sleep 1
done
timeout -k 10 120 python3 -m triton_client_amd.perf -m simple_string -u 127.0.0.1:18101 -i grpc --shared-memory hip \
    --concurrency-range 16:16 --native -p 2000 --stability-windows 2 > "$SUM/string_serving_perf.txt" 2>&1
RC=$?
kill $HOLD 2>/dev/null
wait $SRV
rm -f /tmp/srv_stdin2
tail -8 "$SUM/string_serving_perf.txt"
[ $RC -eq 0 ] || { tail -20 "$OUT/strsrv.err"; exit 1; }
python3 tools/summarize_stats.py "$(find "$OUT/strsrv" -name '*kernel_stats.csv' | head -1)" "$SUM/string_serving_kernel_stats.csv" || exit 1
rm -rf "$OUT/strsrv"
head -12 "$SUM/string_serving_kernel_stats.csv"
