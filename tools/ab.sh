# same-box A/B: _ab_old/ (git worktree of the last commit, built) vs the working tree; alternating runs
# prepare once:  git worktree add -f _ab_old HEAD~0 && (cd _ab_old && bash octave_amd/csrc/build.sh); add _ab_old/ to .git/info/exclude
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
N=${1:-3}
for i in $(seq $N); do
(cd _ab_old && timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | sed 's/^/OLD /') >> gpurun_out/ab.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | sed 's/^/NEW /' >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
