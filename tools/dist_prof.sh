export HSA_ENABLE_IPC_MODE_LEGACY=0 OCTA_DIST_ALWAYS=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
bash tools/stats_step.sh dist --grad-comm bf16 > gpurun_out/stats_dist_summary.txt 2>&1
head -40 gpurun_out/stats_dist_summary.txt | cut -c1-150
