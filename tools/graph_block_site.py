"""Which call of the replayed step blocks the host when several steps are queued?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid
from octave_amd import functional as F_

B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, compute_dtype=torch.bfloat16)
x, ys, real = bench.synth_batch(B, H, 0, dev)
pyr = mask_pyramid(real)
step.capture(x, ys, pyr)
cap = step._caps[H]
g1, g2, g2b, g3 = cap.graphs
names = ["load_static", "refill", "advance", "g1", "g2", "g2b", "g3"]
for trial in range(2):
    torch.cuda.synchronize()
    rows = []
    for k in range(12):
        t = [time.perf_counter()]
        step._load_static(cap, x, ys, pyr); t.append(time.perf_counter())
        cap.feed.refill(); t.append(time.perf_counter())
        step.seg_arena.advance_dyn(step._dyn[0], step.betas); step.disc_arena.advance_dyn(step._dyn[1], step.betas); t.append(time.perf_counter())
        g1.replay(); t.append(time.perf_counter())
        g2.replay(); t.append(time.perf_counter())
        g2b.replay(); t.append(time.perf_counter())
        g3.replay(); t.append(time.perf_counter())
        F_.bump_weight_epoch()
        rows.append([(b - a) * 1e3 for a, b in zip(t, t[1:])])
    torch.cuda.synchronize()
    print("step " + " ".join(f"{n:>11s}" for n in names))
    for k, r in enumerate(rows):
        print(f"{k:4d} " + " ".join(f"{v:11.1f}" for v in r), flush=True)
