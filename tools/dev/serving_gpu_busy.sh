#!/bin/bash
# How busy the GPU is while the server answers the headline load (ResNet-50, batch 8, gRPC + hipshm,
# concurrency 64): the serving process runs under a kernel trace, and the share of wall time in which
# at least one kernel executes (and the mean number executing at once) is computed over the steady part.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
rm -f /tmp/srv_stdin3; mkfifo /tmp/srv_stdin3
( sleep 1000 > /tmp/srv_stdin3 ) &
HOLD=$!
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/busy" -o b -- \
    python3 -m triton_client_amd.server --gpu 0 --models resnet50 --http-port 18200 --grpc-port 18201 \
    --exit-on-stdin-close < /tmp/srv_stdin3 > "$OUT/busy.out" 2>"$OUT/busy.err" &
SRV=$!
for i in $(seq 1 150); do
    python3 - <<'PY' && break
import sys, urllib.request
try:
    sys.exit(0 if urllib.request.urlopen("http://127.0.0.1:18200/v2/models/resnet50/ready", timeout=1).status == 200 else 1)
except Exception:
    sys.exit(1)
PY
    sleep 2
done
TCA_PERF_CONNECTIONS=8 timeout -k 10 120 python3 -m triton_client_amd.perf -m resnet50 -u 127.0.0.1:${PORT:-18201} -i ${PROTO:-grpc} -b 8 --shared-memory ${SHM:-hip} \
    --concurrency-range ${CONC:-64}:${CONC:-64} --native -p 1500 --stability-windows 2 > "$OUT/busy_perf.txt" 2>&1
RC=$?
kill $HOLD 2>/dev/null
wait $SRV
rm -f /tmp/srv_stdin3
tail -3 "$OUT/busy_perf.txt"
[ $RC -eq 0 ] || { tail -20 "$OUT/busy.err"; exit 1; }
python3 - "$(find "$OUT/busy" -name '*kernel_trace.csv' | head -1)" "$SUM/serving_gpu_busy_${SHM:-hip}.txt" <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# the load ran at the end of the trace: the last 2.5 s, minus the final half second
hi = t1 - 500_000_000
lo = hi - 2_000_000_000
sel = [(max(s, lo), min(e, hi)) for s, e, _ in rows if e > lo and s < hi]
busy, cur_s, cur_e, total = 0, None, None, 0
gaps = []
for s, e in sel:
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = hi - lo
gaps.sort()
text = (f"window {wall / 1e6:.0f} ms, {len(sel)} kernels\n"
        f"GPU busy (>= 1 kernel executing): {100.0 * busy / wall:.1f} %\n"
        f"mean kernels executing at once: {total / wall:.2f}\n"
        f"idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.1f} ms, median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
        f"p99 {gaps[int(len(gaps) * 0.99)] / 1e3 if gaps else 0:.1f} us, longest {gaps[-1] / 1e3 if gaps else 0:.1f} us\n")
# time per kernel family over the window (copy kernels: gathers, scatters and the staging of host tensors)
fam = {}
for s, e, name in rows:
    if e > lo and s < hi:
        key = name.split("(")[0].split("<")[0][:40]
        d = fam.setdefault(key, [0, 0])
        d[0] += 1
        d[1] += min(e, hi) - max(s, lo)
for key, (n, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:8]:
    text += f"  {key:40s} {n:7d} launches, {ns / 1e6:8.1f} ms in flight ({100.0 * ns / wall:5.1f} % of the window), mean {ns / n / 1e3:7.1f} us\n"
copies = sorted(e - s for s, e, name in rows if e > lo and s < hi and "BatchedCopy" in name)
if copies:
    pick = lambda q: copies[min(len(copies) - 1, int(len(copies) * q))] / 1e3
    text += (f"copy kernels: {len(copies)} launches, p10 {pick(0.1):.1f} / p50 {pick(0.5):.1f} / p90 {pick(0.9):.1f} / "
             f"max {copies[-1] / 1e3:.1f} us; over 100 us: {sum(1 for c in copies if c > 100_000)}\n")
open(sys.argv[2], "w").write(text)
print(text)
PY
rm -rf "$OUT/busy"
