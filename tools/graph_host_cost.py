"""Host time of every step's launch (hipGraph replay vs eager) over a 20-step run with no synchronisation in between."""
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
batch = (x, ys, mask_pyramid(real))
if len(sys.argv) > 1:
    F_.load_algo_cache(sys.argv[1])
step.capture(*batch)
for mode in ("graph", "eager", "graph"):
    step.launch = mode
    for _ in range(3):
        step(*batch)
    torch.cuda.synchronize()
    ts = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(20):
        a = time.perf_counter()
        step(*batch)
        evs[k + 1].record()
        ts.append((time.perf_counter() - a) * 1e3)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("   device ms between step-end events: " + " ".join(f"{evs[k].elapsed_time(evs[k + 1]):.0f}" for k in range(20)), flush=True)
    print(f"{mode}: total {tot / 20 * 1e3:.1f} ms/step, host {host / 20 * 1e3:.1f} ms/step; per step host ms: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    if mode == "graph":
        # the four replays of one step, individually
        cap = step._caps[H]
        for rep in range(2):
            torch.cuda.synchronize()
            parts = []
            for g in cap.graphs:
                a = time.perf_counter(); g.replay(); parts.append((time.perf_counter() - a) * 1e3)
            print("   replay host ms per graph (GPU idle at start): " + " ".join(f"{p:.2f}" for p in parts), flush=True)
        torch.cuda.synchronize()
