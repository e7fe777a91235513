"""Repeat one stage fwd/bwd parity check N times in one process and print the error spread (debug aid for
order-dependent float-atomic noise vs. real races)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_ops as R
from oracle.fill import fill_state_dict, hash_input
from architectures.models.octa import OctaScribbleNet

name = sys.argv[1] if len(sys.argv) > 1 else "encoder_2"
shape = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 256, 12, 12)
idx = {"encoder_1": 0, "encoder_2": 1, "encoder_3": 2, "encoder_4": 3}[name]
dev = torch.device("cuda:0")
net = OctaScribbleNet(torch.Size((2, 3, 48, 48)), torch.Size((2, 2, 48, 48)), True, False)
fill_state_dict(net.state_dict())
P0 = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to(dev).train()
mod = getattr(net.segmentor, name)
pref = "segmentor." + name
x = hash_input(shape, 77, -1, 1)
Ps = {k: v.clone() for k, v in P0.items() if k.startswith(pref + ".")}
for k, v in Ps.items():
    if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
        v.requires_grad_(True)
xr = x.clone().requires_grad_(True)
out_r = R.encoder_stage(xr, Ps, pref, idx)
cot = hash_input(tuple(out_r.shape), 88, -1, 1)
(out_r * cot).sum().backward()
outs, grads = [], []
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    net.load_state_dict(P0)
    xd = x.to(dev).requires_grad_(True)
    o = mod(xd)
    (o.float() * cot.to(dev)).sum().backward()
    eo = (o.detach().cpu() - out_r.detach()).abs().max().item() / out_r.abs().max().item()
    eg = (xd.grad.cpu() - xr.grad).abs()
    nbad = int((eg > 5e-3 * xr.grad.abs().max()).sum())
    print(f"run {it}: out rel err {eo:.2e}  grad_x rel err {eg.max().item() / xr.grad.abs().max().item():.2e}  elements > 5e-3: {nbad}", flush=True)
    outs.append(o.detach().float().cpu()); grads.append(xd.grad.cpu().clone())
    for p in mod.parameters():
        p.grad = None
print("run-to-run: out max diff", max((outs[0] - o).abs().max().item() for o in outs[1:]), " grad max diff", max((grads[0] - g).abs().max().item() for g in grads[1:]))
