"""Stability of the replayed training step: N steps back to back (no host sync inside a window), a fresh synthetic batch every
window, losses and parameter health per window.  usage: python tools/long_run.py [steps] [window] [dtype: bf16|f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
window = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f16 = len(sys.argv) > 3 and sys.argv[3] == "f16"
B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, lr=1e-4, compute_dtype=torch.float16 if f16 else torch.bfloat16, loss_scale="dynamic" if f16 else 1.0)
x, ys, real = bench.synth_batch(B, H, 0, dev)
step.capture(x, ys, mask_pyramid(real))
step.launch = "graph"
t0 = time.perf_counter()
for w in range(steps // window):
    x, ys, real = bench.synth_batch(B, H, w, dev)
    pyr = mask_pyramid(real)
    for _ in range(window):
        out = step(x, ys, pyr)
    torch.cuda.synchronize()
    bad = sum(int((~torch.isfinite(p)).sum()) for p in net.parameters())
    ls = f" scale {float(step.ls_state[0]):.0f}" if step.ls_state is not None else ""
    print(f"steps {(w + 1) * window:4d}: loss_seg {float(out['loss_seg']):.4f} wpce {float(out['wpce']):.4f} dice {float(out['dice']):.4f} kl {float(out['kl']):.3f} "
          f"g_adv {float(out['g_adv']):.3f} loss_disc {float(out['loss_disc']):.3f} non-finite params {bad}{ls}  ({(time.perf_counter() - t0) / ((w + 1) * window) * 1e3:.1f} ms/step)", flush=True)
    assert bad == 0
print("ok")
