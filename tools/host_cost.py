"""Host-side cost of one replayed step: time each part of TrainStep._replay with the GPU idle (sync before each part)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid

dev = torch.device("cuda", 0)
B, H = 16, 400
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
x, ys, real = synth_batch(B, H, 0, dev)
pyr = mask_pyramid(real)
step.capture(x, ys, pyr)
for _ in range(3):
    step(x, ys, pyr)
torch.cuda.synchronize()
g1, g2, g2b, g3 = step._graphs
acc = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0


def copies():
    step._sx.copy_(x, non_blocking=True); step._sys.copy_(ys, non_blocking=True)
    for d, s in zip(step._sreal, pyr):
        d.copy_(s, non_blocking=True)


N = 10
for _ in range(N):
    timed("input copies", copies)
    timed("rng refill", step._feed.refill)
    timed("advance_dyn x2", lambda: (step.seg_arena.advance_dyn(step._dyn[0], step.betas), step.disc_arena.advance_dyn(step._dyn[1], step.betas)))
    timed("g1 (segmentor fwd/bwd)", g1.replay)
    timed("g2 (discriminator step)", g2.replay)
    timed("g2b (seg Adam+repack)", g2b.replay)
    timed("g3 (disc Adam+repack)", g3.replay)
print(f"{'part':28s} {'host enqueue ms':>16s} {'enqueue+GPU ms':>16s}")
for k, (a, b) in acc.items():
    print(f"{k:28s} {a / N * 1e3:16.2f} {b / N * 1e3:16.2f}")
print(f"{'total':28s} {sum(a for a, _ in acc.values()) / N * 1e3:16.2f} {sum(b for _, b in acc.values()) / N * 1e3:16.2f}")
t0 = time.perf_counter()
for _ in range(N):
    step(x, ys, pyr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"pipelined: host {1e3 * (t1 - t0) / N:.2f} ms/step, wall {1e3 * (t2 - t0) / N:.2f} ms/step; os.cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}")
