# Round profile: bench JSON line, rocprofv3 kernel stats (hipGraph replay) and the two PMC passes (eager, separate runs).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
# (the first bench run measures the per-shape kernel choices and writes them to $OUT/algo_cache.json: no autotune launches in the traces)
python3 bench.py --steps 20 --warmup 3 --algo-cache $OUT/algo_cache.json > $OUT/bench.json 2> $OUT/bench.log
tail -1 $OUT/bench.log
# wall clock on the shared boxes varies run to run (other tenants): two more plain runs for the spread
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-dice --sustained 0 >> $OUT/bench_repeat.json 2>> $OUT/bench.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-dice --sustained 0 >> $OUT/bench_repeat.json 2>> $OUT/bench.log
grep -o 'ms_per_step": [0-9.]*' $OUT/bench.json $OUT/bench_repeat.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --steps 5 --warmup 2 --launch graph --no-cpu-baseline --no-roofline --no-dice --sustained 0 --algo-cache $OUT/algo_cache.json > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --algo-cache $OUT/algo_cache.json > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --algo-cache $OUT/algo_cache.json > $OUT/write.log 2>&1
echo "write pass done"
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json
python3 tools/stats_summary.py $OUT/stats/k_kernel_stats.csv 11 > $OUT/stats_summary.txt
python3 tools/trace_last_step.py $OUT/stats/k_kernel_trace.csv > $OUT/last_step.txt 2>&1 || true
rm -f $OUT/fetch/*kernel_trace.csv $OUT/write/*kernel_trace.csv $OUT/stats/*kernel_trace.csv
ls -la $OUT $OUT/stats | head -30
