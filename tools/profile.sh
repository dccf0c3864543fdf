#!/bin/bash
# GPU-box profiling recipe.  Writes raw traces under gpurun_out/prof and the
# summaries worth keeping under gpurun_out/profiles (copy those into profiles/).
#   1. per-kernel time of every hand-written kernel + the ResNet-50 graph
#   2. LDS / wave counters of the preprocess kernel (own run, counters only)
#   3. per-kernel time inside a serving process under closed-loop gRPC+hipshm load
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"

echo "== 1. kernel trace" && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernels" -o k -- \
    python3 tools/profile_kernels.py > "$SUM/kernels_wall.txt" 2>"$OUT/kernels.err" || { tail -20 "$OUT/kernels.err"; exit 1; }
python3 tools/summarize_stats.py "$(find "$OUT/kernels" -name '*kernel_stats.csv' | head -1)" "$SUM/kernel_stats.csv" || exit 1
rm -rf "$OUT/kernels"
cat "$SUM/kernels_wall.txt"; head -30 "$SUM/kernel_stats.csv"

echo "== 2. preprocess counters" && \
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d "$OUT/pmc" -o p -- python3 tools/profile_kernels.py --only preprocess --iters 3 \
    > "$OUT/pmc.out" 2>"$OUT/pmc.err" || { tail -20 "$OUT/pmc.err"; exit 1; }
python3 - "$OUT/pmc" "$SUM/preprocess_pmc.csv" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = (r.get("Kernel_Name", "?")[:80], r.get("Counter_Name", "?"))
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
with open(sys.argv[2], "w") as out:
    out.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
    for (k, c), (s, n) in sorted(agg.items()):
        out.write(f"\"{k}\",{c},{n},{s:.0f},{s / n:.1f}\n")
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT/pmc"

echo "== 3. serving process under load"
mkfifo /tmp/srv_stdin
( sleep 1000 > /tmp/srv_stdin ) &
HOLD=$!
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/server" -o s -- \
    python3 -m triton_client_amd.server --gpu 0 --models resnet50 --http-port 18000 --grpc-port 18001 \
    --exit-on-stdin-close < /tmp/srv_stdin > "$OUT/server.out" 2>"$OUT/server.err" &
SRV=$!
for i in $(seq 1 150); do
    python3 - <<'PY' && break
import sys, urllib.request
try:
    sys.exit(0 if urllib.request.urlopen("http://127.0.0.1:18000/v2/models/resnet50/ready", timeout=1).status == 200 else 1)
except Exception:
    sys.exit(1)
PY
    sleep 2
done
timeout -k 10 200 python3 -m triton_client_amd.perf -m resnet50 -u 127.0.0.1:18001 -i grpc -b 8 --shared-memory hip \
    --concurrency-range 64:64 --native -p 3000 --stability-windows 2 --json "$SUM/serving_c64_perf.json" > "$SUM/serving_c64_perf.txt" 2>&1
RC=$?
kill $HOLD 2>/dev/null
wait $SRV
rm -f /tmp/srv_stdin
tail -15 "$SUM/serving_c64_perf.txt"
[ $RC -eq 0 ] || exit 1
python3 tools/summarize_stats.py "$(find "$OUT/server" -name '*kernel_stats.csv' | head -1)" "$SUM/serving_kernel_stats.csv" || exit 1
rm -rf "$OUT/server"
head -30 "$SUM/serving_kernel_stats.csv"
