"""What will the gradient exchange's kernels cost the backward pass they run beside?  One GPU cannot run an 8-rank ring, but it can run a
stand-in with the same footprint: with a single-rank RCCL communicator (OCTA_DIST_ALWAYS=1: broadcast, piecewise graphs, bucket hand-over
at the stage marks, all as at N > 1) every bucket's all-reduce on the comm stream is followed by `nblocks` workgroups that stream over the
bucket for as long as a ring all-reduce of that bucket would take at `gbs` GB/s of algorithmic bandwidth (octa_probe_stream_load).  The
replayed step is timed for several footprints; the delta against nblocks = 0 is the price of sharing CUs / HBM with the exchange.
usage (GPU box): OCTA_DIST_ALWAYS=1 python tools/comm_pressure.py > profiles/r05_comm_pressure.txt"""
import os, sys, time
os.environ.setdefault("OCTA_DIST_ALWAYS", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from bench import synth_batch
from architectures.models.octa import OctaScribbleNet
from octave_amd import train as T
from octave_amd._lib import lib
from octave_amd.train import TrainStep, mask_pyramid

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
B, H = 16, 400
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
x, ys, real = synth_batch(B, H, 0, dev)
pyr = mask_pyramid(real)
PROBE = {"nblocks": 0, "gbs": 200.0}
orig = T.FlatArena._reduce_range


def reduce_and_probe(self, lo, hi, comm_dtype):
    orig(self, lo, hi, comm_dtype)
    nb = PROBE["nblocks"]
    if nb > 0 and self.g.is_cuda:
        nbytes = (hi - lo) * (2 if comm_dtype == torch.bfloat16 else 4)
        # one pass of nb workgroups over the bucket moves 2 x nbytes; a workgroup streams ~25 GB/s, so reps makes the kernel last as long
        # as the ring would at `gbs` GB/s of algorithmic bandwidth
        t_ring = nbytes / (PROBE["gbs"] * 1e9)
        t_pass = 2.0 * nbytes / (nb * 25e9)
        reps = max(1, int(round(t_ring / t_pass)))
        lib().octa_probe_stream_load(self.g.data_ptr() + 4 * lo, (hi - lo) * 4, nb, reps, torch.cuda.current_stream().cuda_stream)


T.FlatArena._reduce_range = reduce_and_probe
step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16, grad_comm_dtype=torch.bfloat16)
step.capture(x, ys, pyr)
print(f"single-rank RCCL path, bf16 exchange, buckets: {[(t, (hi - lo) * 2 >> 20) for t, lo, hi in step.seg_arena.buckets]} (MB)")
for rnd in range(3):
    for nb, gbs in ((0, 200.0), (16, 200.0), (32, 200.0), (64, 200.0), (32, 100.0)):
        PROBE["nblocks"], PROBE["gbs"] = nb, gbs
        for _ in range(5):
            step(x, ys, pyr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step(x, ys, pyr)
        torch.cuda.synchronize()
        print(f"round {rnd}: comm-stream stand-in {nb:3d} workgroups for the duration of a ring at {gbs:5.0f} GB/s algbw: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
dist.destroy_process_group()
