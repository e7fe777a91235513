# PMC passes over the wgrad micro-benchmark (one layer): SQ utilisation, L2 hit rate, HBM fetch.
# usage (GPU box, repo root): bash tools/pmc_wg8.sh <layer> <tag>
set -e
LAYER=${1:-dec2_3x3}
TAG=${2:-a}
R=$PWD
OUT=$R/gpurun_out/pmc_${LAYER}_$TAG
mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 $R/tools/wgrad_micro.py time $LAYER > $OUT/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- python3 $R/tools/wgrad_micro.py time $LAYER > $OUT/tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/tools/wgrad_micro.py time $LAYER > $OUT/fetch.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/lds -o p -- python3 $R/tools/wgrad_micro.py time $LAYER > $OUT/lds.log 2>&1
cd $R
python3 - <<PY
import csv, collections, glob
for sub in ("sq", "tcc", "fetch", "lds"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "wgrad" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(sub, k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
find $OUT -name "*kernel_trace.csv" -delete
