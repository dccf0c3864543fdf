#!/bin/bash
# Regenerates every summary kept under profiles/ on a GPU box (writes gpurun_out/profiles/).
# Steps stop at the first failure; nothing here retries a GPU step.
set -u
SUM=$(pwd)/gpurun_out/profiles
mkdir -p "$SUM"
step() { echo "=== $*"; }
step "kernels + preprocess counters + serving under rocprofv3" && bash tools/profile.sh > "$SUM/profile_sh.log" 2>&1 || { tail -20 "$SUM/profile_sh.log"; exit 1; }
step "resnet kernel stats" && bash tools/profile_resnet.sh > /dev/null 2>&1 && bash tools/profile_resnet_b128.sh 128 > /dev/null 2>&1 && bash tools/profile_resnet_b128.sh 8 > /dev/null 2>&1 || exit 1
step "graph gaps" && bash tools/dev/graph_gap_analysis.sh 8 > /dev/null 2>&1 && bash tools/dev/graph_gap_analysis.sh 128 > /dev/null 2>&1 || exit 1
step "markers" && bash tools/marker_check.sh > /dev/null 2>&1 || exit 1
step "forward time, own kernels vs library" && timeout -k 10 300 python3 tools/model_bench.py > "$SUM/resnet50_forward_fused.txt" 2>&1 && \
    timeout -k 10 300 python3 tools/model_bench.py --no-fused > "$SUM/resnet50_forward_library.txt" 2>&1 || exit 1
step "per-layer convolution bench" && timeout -k 10 600 python3 tools/conv_bench.py > "$SUM/conv_layer_bench.txt" 2>&1 || exit 1
step "split-K bench" && timeout -k 10 300 python3 tools/dev/splitk_bench.py > "$SUM/splitk_bench.txt" 2>&1 || exit 1
step "fused block bench" && timeout -k 10 300 python3 tools/dev/fused_block_bench.py > "$SUM/fused_block_bench.txt" 2>&1 || exit 1
step "conv counters" && bash tools/conv_pmc.sh > /dev/null 2>&1 || exit 1
step "secondary configurations" && timeout -k 10 300 python3 tools/secondary_bench.py --gpu 0 --out "$SUM/secondary_bench.jsonl" > /dev/null 2>&1 || exit 1
step "transport matrix" && timeout -k 10 300 python3 tools/dev/shm_mode_matrix.py 2>&1 | grep RESULT > "$SUM/shm_mode_matrix.txt" || exit 1
step "serving sweep" && timeout -k 10 600 python3 tools/serving_sweep.py --out "$SUM" > "$SUM/serving_sweep.txt" 2>&1 || exit 1
step "bench.py" && timeout -k 10 300 python3 bench.py > "$SUM/bench_c64.log.json" 2>"$SUM/bench.err" || exit 1
tail -1 "$SUM/bench_c64.log.json"
rm -rf gpurun_out/prof
