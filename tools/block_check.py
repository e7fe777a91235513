import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_ops as R
from oracle.fill import fill_state_dict, hash_input
from architectures.models.octa import OctaScribbleNet
dev = torch.device("cuda:0")
net = OctaScribbleNet(torch.Size((2, 3, 48, 48)), torch.Size((2, 2, 48, 48)), True, False)
fill_state_dict(net.state_dict())
P0 = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for stage, idx, shapes in (("encoder_2", 1, [(B, 256, 12, 12)] + [(B, 512, 6, 6)] * 3), ("encoder_1", 0, [(B, 64, 12, 12)] + [(B, 256, 12, 12)] * 2)):
    for bi, shape in enumerate(shapes):
        mod = getattr(net.segmentor, stage)[bi]
        pref = f"segmentor.{stage}.{bi}"
        Ps = {k: v.clone() for k, v in P0.items() if k.startswith(pref + ".")}
        for k, v in Ps.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        x = hash_input(shape, 77 + bi, -1, 1)
        xr = x.clone().requires_grad_(True)
        o_r = R.bottleneck(xr, Ps, pref, 2 if (bi == 0 and idx > 0) else 1, bi == 0)
        cot = hash_input(tuple(o_r.shape), 88, -1, 1)
        (o_r * cot).sum().backward()
        xd = x.to(dev).requires_grad_(True)
        o = mod(xd)
        (o.float() * cot.to(dev)).sum().backward()
        eo = (o.detach().cpu() - o_r.detach()).abs().max().item() / o_r.abs().max().item()
        eg = (xd.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
        worst = ("", 0.0)
        for k, pm in mod.named_parameters():
            w = Ps[pref + "." + k].grad
            if w is None or k.endswith(("fc1.bias", "conv2.conv.bias")):
                continue
            e = (pm.grad.cpu() - w).abs().max().item() / (w.abs().max().item() + 1e-12)
            if e > worst[1]:
                worst = (k, e)
        print(f"{stage}[{bi}] {shape}: out {eo:.2e} grad_x {eg:.2e} worst param grad {worst[0]} {worst[1]:.2e}", flush=True)
