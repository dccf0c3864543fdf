#!/bin/bash
# Hardware counters of the stage-1 bottleneck kernels (row-organised vs chunked), one counter pass per run.
# usage: tools/dev/rows_pmc.sh [batch] [n3] [proj]   -> gpurun_out/r3/rows_pmc.csv
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/r3
mkdir -p "$OUT" "$SUM"
BATCH=${1:-128}; N3=${2:-64}; PROJ=${3:-0}
P1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"
P4="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
P5="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum"
P6="GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum"
n=0
for kind in rows chunked; do
  for pass in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
    n=$((n + 1))
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d "$OUT/rpmc$n" -o p -- \
        python3 tools/dev/rows_pmc.py $kind $BATCH $N3 $PROJ > "$OUT/rpmc$n.out" 2>"$OUT/rpmc$n.err" || { tail -5 "$OUT/rpmc$n.err" | cut -c1-300; exit 1; }
  done
done
python3 - "$OUT" "$SUM/rows_pmc_b${BATCH}_n${N3}_p${PROJ}.csv" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[2], "w")
out.write("kernel,counter,dispatches,mean_per_dispatch\n")
agg = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(sys.argv[1] + "/rpmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "?")
            if "BottleneckRows" in name:
                short = "rows"
            elif "ConvGemm" in name:
                short = "chunked"
            else:
                continue
            k = (short, r.get("Counter_Name", "?"))
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (k, c), (s, cnt) in sorted(agg.items()):
    out.write(f"{k},{c},{cnt},{s / cnt:.1f}\n")
out.close()
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT"/rpmc*
