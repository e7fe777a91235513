"""Where the N > 1 launch path spends its extra time, measured on ONE rank (OCTA_DIST_ALWAYS=1 makes a single-rank RCCL group go
through the bucketed exchange): the full path, the same with dist.all_reduce replaced by a no-op (casts, streams and events stay),
and the same without the bf16 pack.  Usage (GPU box): python tools/dist_overhead_probe.py"""
import os
import sys
import time

os.environ.setdefault("OCTA_DIST_ALWAYS", "1")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from architectures.models.octa import OctaScribbleNet  # noqa: E402
from octave_amd import functional as F_  # noqa: E402
from octave_amd.train import TrainStep, mask_pyramid  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": dev} if os.environ.get("PROBE_DEVICE_ID", "1") == "1" else {}))
B, H = 16, 400
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
x, ys, real = F_.synth_octa_batch(B, H, H, seed=1234, device=dev)
batch = (x, ys, mask_pyramid(real))


def run(tag, comm_dtype, noop):
    real_ar = dist.all_reduce
    if noop:
        dist.all_reduce = lambda *a, **k: None
    try:
        step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16, grad_comm_dtype=comm_dtype)
        step.capture(*batch)
        for _ in range(5):
            step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step(*batch)
        torch.cuda.synchronize()
        print(f"{tag:58s} {(time.perf_counter() - t0) / 30 * 1e3:6.2f} ms/step", flush=True)
        step.close()
    finally:
        dist.all_reduce = real_ar


run("bucketed exchange, bf16 pack, RCCL all-reduce", torch.bfloat16, False)
run("bucketed exchange, bf16 pack, all-reduce = no-op", torch.bfloat16, True)
run("bucketed exchange, fp32, RCCL all-reduce", None, False)
run("bucketed exchange, fp32, all-reduce = no-op", None, True)
dist.destroy_process_group()
