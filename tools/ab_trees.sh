# Same-box A/B of two source TREES (e.g. _ab_old/ = `git archive HEAD` built in place, and the working tree) on the replayed step, alternating.
# usage (GPU box): bash tools/ab_trees.sh ROUNDS TREE_A TREE_B        (a tree = a directory with bench.py and a built octave_amd/libocta_hip.so)
N=$1; A=$2; B=$3
mkdir -p gpurun_out
for t in $A $B; do python $t/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph --algo-cache gpurun_out/ab_trees_cache_$(basename $t).json > /dev/null 2>&1; done
for i in $(seq $N); do
  for t in $A $B; do
    python $t/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph --algo-cache gpurun_out/ab_trees_cache_$(basename $t).json 2>&1 | grep "timed region" | sed "s|^|[$t] |" | cut -c1-110
  done
done
