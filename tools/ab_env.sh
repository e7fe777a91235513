# Same-box A/B of two environment settings (e.g. kernel families on / off): wall clock per step of the replayed step, alternating.
# usage (GPU box): bash tools/ab_env.sh "OCTA_NO_HALO8=1 OCTA_NO_PWGEMM=1" "" [rounds]      (every run autotunes its own kernel choices)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph 2>&1 | grep "timed region" | sed "s|^|[$E] |" | cut -c1-140
  done
done
