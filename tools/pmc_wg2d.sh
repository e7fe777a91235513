# PMC passes (L2 hit rate, HBM fetch, LDS) over one 3x3 layer's weight gradient on wgrad9 (mode 0) and wgrad2d (mode 1).
# usage (GPU box, repo root): bash tools/pmc_wg2d.sh <layer>
LAYER=${1:-d3_3x3}
R=$PWD
OUT=$R/gpurun_out/pmc_wg2d_$LAYER
mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
cd /tmp
for m in 0 1; do
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/tcc$m -o p -- python3 $R/tools/wgrad2d_one.py $LAYER $m > $OUT/tcc$m.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $OUT/fetch$m -o p -- python3 $R/tools/wgrad2d_one.py $LAYER $m > $OUT/fetch$m.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/lds$m -o p -- python3 $R/tools/wgrad2d_one.py $LAYER $m > $OUT/lds$m.log 2>&1
echo "mode $m done"
done
cd $R
python3 - <<PY
import csv, collections, glob
for sub in ("tcc0", "tcc1", "fetch0", "fetch1", "lds0", "lds1"):
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
