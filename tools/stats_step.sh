# rocprofv3 kernel stats of the replayed training step -> gpurun_out/stats_<tag>/ (+ a per-family summary)
# usage (GPU box, repo root): bash tools/stats_step.sh <tag> [extra bench args]
TAG=${1:-x}; shift
R=$PWD
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $R/bench.py --steps 5 --warmup 2 --launch graph --no-cpu-baseline --no-roofline --no-dice --sustained 0 "$@" > $OUT/run.log 2>&1
cd $R
python3 tools/trace_last_step.py $OUT/k_kernel_trace.csv > $OUT/last_step.txt 2>&1
rm -f $OUT/*kernel_trace.csv
python3 tools/stats_summary.py $OUT/k_kernel_stats.csv 9
