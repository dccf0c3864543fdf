#!/bin/bash
# Per-kernel time of the ResNet-50 graph alone (b8, b64, b128 replays).
set -u
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof
SUM=$(pwd)/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/resnet" -o r -- \
    python3 tools/profile_kernels.py --only resnet > "$SUM/resnet_wall.txt" 2>"$OUT/resnet.err" || { tail -20 "$OUT/resnet.err"; exit 1; }
python3 tools/summarize_stats.py "$(find "$OUT/resnet" -name '*kernel_stats.csv' | head -1)" "$SUM/resnet_kernel_stats.csv" 30 || exit 1
rm -rf "$OUT/resnet"
cat "$SUM/resnet_wall.txt"; cut -c1-150 "$SUM/resnet_kernel_stats.csv"
