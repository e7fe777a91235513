#!/bin/bash
# PMC counters of one conv kernel: tools/pmc_conv.sh OUTDIR LAYER ALGO|wgrad [fwd|dgrad]   (run on the GPU box; separate passes per counter group)
set -u
OUT=$1; LAYER=$2; ALGO=$3; KIND=${4:-fwd}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  if [ "$ALGO" = "wgrad" ]; then
    rocprofv3 --pmc $grp --kernel-trace -d "$R/$OUT/p$i" -o pmc -f csv -- python3 "$R/tools/wgrad_one.py" "$LAYER" 3 > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$R/$OUT/fail.log"
  else
    rocprofv3 --pmc $grp --kernel-trace -d "$R/$OUT/p$i" -o pmc -f csv -- python3 "$R/tools/conv_one.py" "$LAYER" "$ALGO" "$KIND" 3 > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$R/$OUT/fail.log"
  fi
done
python3 - "$R/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "halo8" in k or "igemm8" in k or "pwgemm" in k or "wgrad9" in k or "wgrad8" in k or "conv_res" in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:32s} {sum(v) / len(v):16.0f}   (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
