"""How far do the fp32 gradient norms of the 64^2 golden network move from run to run (float atomics in the weight
gradients + chaotic BatchNorms over 12 samples)?  Prints worst deviation from the float64 fixture and the parameter.
usage: python tools/grad_flake.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_hip_modules import _build, _scribble, hash_input
from conftest import load_golden
from architectures.segmentor.losses import DiceLoss

dev = torch.device("cuda:0")
Hn, Bn = 64, 3
G = load_golden(f"unet_{Hn}.npz")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1).to(dev)
ys = _scribble(Bn, Hn).to(dev)
prev = None
for it in range(runs):
    net, _ = _build(Bn, Hn, dev)
    att, agg, x4 = net.segmentor(x)
    p = torch.softmax(agg, dim=1)
    loss = net.supervised_loss(p, ys) + DiceLoss()(p, ys)
    loss.backward()
    params = dict(net.segmentor.named_parameters())
    devs = {}
    for k, g in G.items():
        if k.startswith("gradnorm_f64/"):
            name = k[len("gradnorm_f64/"):]
            g64 = float(g)
            if g64 < 1e-9:
                continue
            devs[name] = (abs(params[name].grad.double().norm().item() - g64) / g64, abs(float(G["gradnorm/" + name]) - g64) / g64, g64)
    worst = sorted(devs.items(), key=lambda kv: -kv[1][0])[:3]
    cur = torch.cat([params[n].grad.flatten() for n in sorted(devs)])
    same = "" if prev is None else f" | max|grad - previous run| {float((cur - prev).abs().max()):.3e}"
    prev = cur
    print(f"run {it}: loss {loss.item():.6f} worst " + "; ".join(f"{n} hip {d[0]:.3f} ref {d[1]:.3f} |g64| {d[2]:.2e}" for n, d in worst) + same, flush=True)
