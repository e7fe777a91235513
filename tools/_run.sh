set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1 || { tail -30 gpurun_out/t_gpu.log; exit 1; }
tail -2 gpurun_out/t_gpu.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>gpurun_out/err_new.log | cut -c1-160
done
tail -1 gpurun_out/err_new.log
