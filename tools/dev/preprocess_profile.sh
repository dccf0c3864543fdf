#!/bin/bash
# Preprocess kernel on the GPU box: numerics tests, per-kernel time with and without the staged
# source tile, LDS counters (own run).  Summaries land in gpurun_out/profiles/.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
timeout -k 10 240 python3 -u -m pytest tests/test_gpu_kernels.py -k preprocess -x -q --timeout 120 --timeout-method thread || exit 1
for stage in 1 0; do
  export TCA_PREPROCESS_STAGE=$stage
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pk$stage" -o k -- \
      python3 tools/profile_kernels.py --only preprocess --iters 20 > "$OUT/pk$stage.out" 2>"$OUT/pk$stage.err" || { tail -20 "$OUT/pk$stage.err"; exit 1; }
  python3 tools/summarize_stats.py "$(find "$OUT/pk$stage" -name '*kernel_stats.csv' | head -1)" "$SUM/preprocess_kernel_stats_stage$stage.csv" || exit 1
  rm -rf "$OUT/pk$stage"
  echo "== stage=$stage"; grep -i preprocess "$SUM/preprocess_kernel_stats_stage$stage.csv"
done
export TCA_PREPROCESS_STAGE=1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY \
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
