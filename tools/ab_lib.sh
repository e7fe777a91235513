# Same-box A/B of two builds of the library: wall clock per step of the replayed step, alternating, 30 steps each.
# usage (GPU box): bash tools/ab_lib.sh <base.so> <new.so> [rounds]
A=$1; B=$2; N=${3:-3}
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --algo-cache gpurun_out/ab_cache.json > /dev/null 2>&1
for i in $(seq $N); do
  for L in $A $B; do
    OCTA_HIP_LIB=$PWD/$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --launch graph --algo-cache gpurun_out/ab_cache.json 2>&1 | grep "timed region" | sed "s|^|$L |" | cut -c1-110
  done
done
