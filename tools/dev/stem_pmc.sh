#!/bin/bash
# Hardware counters of the fused stem kernel at batch 128 and 8, one rocprofv3 pass per counter set.
# Writes gpurun_out/profiles/stem_pmc.csv.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/stem_pmc_raw
SUM=gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
PASS1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PASS2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
n=0
for batch in 128 8; do
  for pass in "$PASS1" "$PASS2"; do
    n=$((n + 1))
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d "$OUT/s$n" -o p -- \
        python3 tools/dev/stem_pmc.py --batch $batch > "$OUT/s$n.out" 2>"$OUT/s$n.err" || { grep -a -A8 Traceback "$OUT/s$n.err" | cut -c1-300; exit 1; }
    tail -1 "$OUT/s$n.out"
  done
done
python3 - "$OUT" "$SUM/stem_pmc.csv" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[2], "w")
out.write("run,what,counter,dispatches,mean_per_dispatch\n")
for d in sorted(glob.glob(sys.argv[1] + "/s*/")):
    run = d.rstrip("/").split("/")[-1]
    what = open(d.rstrip("/") + ".out").read().strip().splitlines()[-1].split(":")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "StemConvPool" not in r.get("Kernel_Name", ""):
                continue
            k = r.get("Counter_Name", "?")
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
    for c, (s, cnt) in sorted(agg.items()):
        out.write(f"{run},{what},{c},{cnt},{s / cnt:.1f}\n")
out.close()
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT"
