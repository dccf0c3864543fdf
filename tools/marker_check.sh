#!/bin/bash
# ROCTx ranges next to the kernels they enclose: a short ResNet-50 graph-replay
# run under rocprofv3 with marker + kernel tracing (no counters in this run).
set -u
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof/markers
SUM=$(pwd)/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TCA_ROCTX=1
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OUT" -o m -- \
    python3 tools/model_bench.py --batches 8,128 --iters 20 > "$OUT/run.log" 2>"$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }

python3 - "$OUT" "$SUM/marker_ranges.csv" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*marker_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Name") or "?"
        try:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        except Exception:
            dur = 0.0
        agg[name][0] += 1
        agg[name][1] += dur
with open(sys.argv[2], "w") as out:
    out.write("range,count,mean_us\n")
    for name, (n, total) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f"\"{name[:80]}\",{n},{total / max(n, 1):.1f}\n")
print(open(sys.argv[2]).read()[:1500])
PY
python3 - "$OUT" "$SUM/kernel_gaps.txt" <<'PY'
# idle time between consecutive kernels of a graph replay (same queue, gaps < 100 us)
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
gaps, busy = [], 0
for (s0, e0, _), (s1, e1, _) in zip(rows, rows[1:]):
    g = s1 - e0
    if 0 <= g < 100000:
        gaps.append(g)
busy = sum(e - s for s, e, _ in rows)
gaps.sort()
with open(sys.argv[2], "w") as out:
    if gaps:
        out.write(f"kernels {len(rows)}  busy {busy / 1e6:.2f} ms  gaps {len(gaps)}  idle-in-gaps {sum(gaps) / 1e6:.2f} ms\n")
        out.write(f"gap us: mean {sum(gaps) / len(gaps) / 1e3:.2f}  p50 {gaps[len(gaps) // 2] / 1e3:.2f}  "
                  f"p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f}  max {gaps[-1] / 1e3:.2f}\n")
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT"
