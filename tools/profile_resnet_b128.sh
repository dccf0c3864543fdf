#!/bin/bash
# Per-kernel time of ONE ResNet-50 batch size (default 128): where a forward's time goes.
set -u
export TMPDIR=/tmp
B=${1:-128}
OUT=$(pwd)/gpurun_out/prof
SUM=$(pwd)/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/resnet_b" -o r -- \
    python3 tools/profile_kernels.py --only resnet --resnet-batch-only --batch "$B" --iters 40 > "$SUM/resnet_b${B}_wall.txt" 2>"$OUT/resnet_b.err" || { tail -20 "$OUT/resnet_b.err"; exit 1; }
python3 tools/summarize_stats.py "$(find "$OUT/resnet_b" -name '*kernel_stats.csv' | head -1)" "$SUM/resnet_b${B}_kernel_stats.csv" 30 || exit 1
rm -rf "$OUT/resnet_b"
cat "$SUM/resnet_b${B}_wall.txt"; cut -c1-150 "$SUM/resnet_b${B}_kernel_stats.csv"
