#!/bin/bash
# Hardware counters of the fused bottleneck-block kernel (stage-1 shape, batch 128):
# where the waves' time goes (SQ), and what reaches L2 / memory (TCC).  One pass per run.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
LAYER="${FUSED_LAYER:---hw 56 --cin 64 --cout 64 --ksize 3 --expand 256 --next 64}"
PASSES=("SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
        "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
        "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
        "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE")
n=0
for pass in "${PASSES[@]}"; do
  n=$((n + 1))
  timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d "$OUT/fpmc$n" -o p -- \
      python3 tools/conv_pmc.py --batch 128 $LAYER > "$OUT/fpmc$n.out" 2>"$OUT/fpmc$n.err" || { echo "pass $n failed"; tail -3 "$OUT/fpmc$n.err" | cut -c1-300; continue; }
done
python3 - "$OUT" "$SUM/${FUSED_TAG:-fused_pmc}.csv" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[2], "w")
out.write("run,kernel,counter,dispatches,mean_per_dispatch\n")
for d in sorted(glob.glob(sys.argv[1] + "/fpmc*/")):
    run = d.rstrip("/").split("/")[-1]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "?")
            if "ConvGemm" not in name:
                continue
            k = (name[name.find("ConvGemmKernel"):][:70], r.get("Counter_Name", "?"))
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
    for (k, c), (s, cnt) in sorted(agg.items()):
        out.write(f"{run},\"{k}\",{c},{cnt},{s / cnt:.1f}\n")
out.close()
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT"/fpmc*
