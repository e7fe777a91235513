set -e
R=$PWD; OUT=$R/gpurun_out/r05u; mkdir -p $OUT; export TMPDIR=/tmp
python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --algo-cache $OUT/algo_cache.json > $OUT/b0.json 2> $OUT/b0.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --steps 5 --warmup 2 --launch graph --no-cpu-baseline --no-roofline --no-dice --sustained 0 --algo-cache $OUT/algo_cache.json > $OUT/stats.log 2>&1
cd $R
python3 tools/trace_last_step.py $OUT/stats/k_kernel_trace.csv > $OUT/last_step.txt
rm -f $OUT/stats/*kernel_trace.csv
grep -c . $OUT/last_step.txt
