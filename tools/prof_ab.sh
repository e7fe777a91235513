# kernel-time A/B (robust against host/box noise): rocprofv3 kernel trace of _ab_old/ and of the working tree
# prepare once:  git worktree add -f _ab_old HEAD~0 && (cd _ab_old && bash octave_amd/csrc/build.sh); add _ab_old/ to .git/info/exclude
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pab; rm -rf $R/gpurun_out/pab/*
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/pab -o old -- python3 $R/_ab_old/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/pab/old.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/pab -o new -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/pab/new.log 2>&1
grep -o 'ms_per_step": [0-9.]*' $R/gpurun_out/pab/old.log $R/gpurun_out/pab/new.log
