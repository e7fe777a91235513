run() { env HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 5 --no-dice --sustained 0 --no-cpu-baseline --no-roofline $EXTRA 2>&1 | grep "timed region" | sed "s|^|[$* $EXTRA] |" | cut -c1-150; }
EXTRA="--grad-comm bf16" run OCTA_DIST_ALWAYS=1
EXTRA="--grad-comm f32" run OCTA_DIST_ALWAYS=1
EXTRA="" run OCTA_SPLIT_BACKWARD=1
EXTRA="" run OCTA_X=1
EXTRA="--grad-comm bf16" run OCTA_DIST_ALWAYS=1
EXTRA="" run OCTA_X=1
