#!/bin/bash
# Hardware counters of the MFMA convolution kernel on two compute-heavy ResNet-50
# layers (counters only, one pass per run; <= 8 SQ counters per pass).
# Writes gpurun_out/profiles/conv_pmc.csv.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
PASS1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PASS2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
n=0
for layer in "--hw 14 --cin 256 --cout 256 --ksize 3 --variant 128x128w8" "--hw 14 --cin 1024 --cout 256 --ksize 1 --variant 128x128w8" \
             "--hw 14 --cin 256 --cout 256 --ksize 3 --variant 256x256w16" "--hw 14 --cin 1024 --cout 256 --ksize 1 --variant 256x256w16" \
             "--hw 56 --cin 64 --cout 256 --ksize 1 --variant 128x128w8" "--hw 14 --cin 256 --cout 256 --ksize 3 --variant 128x64" \
             "--hw 56 --cin 64 --cout 64 --ksize 3 --expand 256 --next 64" "--hw 28 --cin 128 --cout 128 --ksize 3 --expand 512"; do
  for pass in "$PASS1" "$PASS2"; do
    n=$((n + 1))
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d "$OUT/cpmc$n" -o p -- \
        python3 tools/conv_pmc.py --batch 128 $layer > "$OUT/cpmc$n.out" 2>"$OUT/cpmc$n.err" || { grep -a -A8 Traceback "$OUT/cpmc$n.err" | cut -c1-300; exit 1; }
    cat "$OUT/cpmc$n.out" | tail -1
  done
done
python3 - "$OUT" "$SUM/conv_pmc.csv" <<'PY'
import csv, glob, sys, collections, re
out = open(sys.argv[2], "w")
out.write("run,layer,kernel,counter,dispatches,mean_per_dispatch\n")
for d in sorted(glob.glob(sys.argv[1] + "/cpmc*/")):
    run = d.rstrip("/").split("/")[-1]
    layer = open(d.rstrip("/") + ".out").read().strip().splitlines()[-1].split(":")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "?")
            if "ConvGemm" not in name:
                continue
            m = re.search(r"ConvGemmKernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)", name)
            wm = re.search(r"ELb\dELi(\d+)EE", name)  # wave rows: the last template argument
            rows = int(wm.group(1)) if wm else 2
            short = (f"ConvGemm<{m.group(1)},{m.group(2)},waves={rows * int(m.group(3))},spatial={m.group(4)},glds={m.group(5)},"
                     f"fused={m.group(6)},next={m.group(7)}>" if m else name[:60])
            k = (short, r.get("Counter_Name", "?"))
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
    for (k, c), (s, cnt) in sorted(agg.items()):
        out.write(f"{run},{layer},\"{k}\",{c},{cnt},{s / cnt:.1f}\n")
out.close()
print(open(sys.argv[2]).read())
PY
rm -rf "$OUT"/cpmc*
