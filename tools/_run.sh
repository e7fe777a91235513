set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "colsum or conv_transpose or conv2d_fwd" > gpurun_out/t_ops.log 2>&1 || { tail -30 gpurun_out/t_ops.log; exit 1; }
tail -2 gpurun_out/t_ops.log
bash tools/prof_ab.sh
