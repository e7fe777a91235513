"""Wall time of consecutive windows of replayed steps (is a fresh box / first process slower, and for how long?)."""
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
torch.cuda.synchronize()
out = []
for w in range(12):
    t0 = time.perf_counter()
    for _ in range(10):
        step(x, ys, pyr)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 10 * 1e3)
print("ms/step per window of 10:", " ".join(f"{v:.1f}" for v in out))
