#!/bin/bash
# How much of a ResNet-50 graph replay is kernel time and how much is the gap between
# dependent kernels: kernel trace of batch $1 (default 8), summarised from the raw trace.
set -u
export TMPDIR=/tmp
B=${1:-8}
OUT=$(pwd)/gpurun_out/prof
mkdir -p "$OUT" gpurun_out/profiles
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/gap" -o g -- \
    python3 tools/profile_kernels.py --only resnet --resnet-batch-only --batch "$B" --iters 40 > "$OUT/gap_wall.txt" 2>"$OUT/gap.err" || { tail -20 "$OUT/gap.err"; exit 1; }
python3 - "$(find "$OUT/gap" -name '*kernel_trace.csv' | head -1)" "$B" <<'PY' | tee gpurun_out/profiles/graph_gaps_b$B.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the last 10 replays: find the period by the stem kernel
stems = [i for i, k in enumerate(ks) if "StemConvPool" in k[2]]
per = stems[-1] - stems[-2]
lo = stems[-11]
seg = ks[lo:stems[-1]]
busy = sum(e - s for s, e, _ in seg)
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1) if (i + 1) % per != 0]
span = sum(seg[(j + 1) * per - 1][1] - seg[j * per][0] for j in range(10))
print(f"batch {sys.argv[2]}: {per} kernels per replay; per replay: span {span / 10 / 1e3:.1f} us, kernel time {busy / 10 / 1e3:.1f} us, "
      f"gaps {sum(gaps) / 10 / 1e3:.1f} us (mean {sum(gaps) / len(gaps) / 1e3:.2f} us, max {max(gaps) / 1e3:.2f} us)")
PY
cat "$OUT/gap_wall.txt"
rm -rf "$OUT/gap"
