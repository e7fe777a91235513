# Same-box A/B of N environment settings on the replayed step, alternating; the conv kernel choices are tuned once and shared
# through an algo cache.  usage (GPU box): bash tools/ab_envs.sh ROUNDS "ENV_A" "ENV_B" ...      ("" = defaults)
N=$1; shift
CACHE=gpurun_out/ab_envs_algo_cache.json
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph --algo-cache $CACHE > /dev/null 2>&1
for i in $(seq $N); do
  for E in "$@"; do
    env $E python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph --algo-cache $CACHE 2>&1 | grep "timed region" | sed "s|^|[$E] |" | cut -c1-150
  done
done
