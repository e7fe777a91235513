"""Debug: after capture + N replays, compare every cached packed operand with a fresh pack of the current weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.fill import fill_state_dict
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid
from octave_amd import functional as F_

dev = torch.device("cuda:0")
Bn, H = 6, 48
g = torch.Generator().manual_seed(5)
x = torch.rand(Bn, 3, H, H, generator=g).to(dev)
ys = (torch.rand(Bn, 2, H, H, generator=g) > 0.7).float().to(dev)
real = (torch.rand(Bn, 2, H, H, generator=g) > 0.5).float().to(dev)
pyr = mask_pyramid(real)
net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
fill_state_dict(net.state_dict())
net = net.to(dev).train()
st = TrainStep(net, lr=1e-4, compute_dtype=torch.float32)
torch.manual_seed(11)
st.capture(x, ys, pyr, warmup=1)
names = {id(p): n for n, p in net.named_parameters()}
for it in range(2):
    stale = []
    for key, e in list(F_._PACK_CACHE.items()):
        w = e.wref()
        if w is None or e.direct:
            continue
        cur = e.out.clone()
        e2tag = e.tag
        e.tag = (None, None) + tuple(e.tag[2:])
        fresh = F_._packed(w, e.kind, e.dtype, e.groups, e.pad_to).clone()   # re-packs in place: same buffer
        d = (cur.float() - fresh.float()).abs().max().item()
        if d > 0:
            stale.append((names.get(key[0], "?"), e.kind, d))
    print(f"after {it} replays: {len(stale)} stale packed operands of {len(F_._PACK_CACHE)}", stale[:8])
    out = st(x, ys, pyr)
    torch.cuda.synchronize()
    print({k: round(v.item(), 4) for k, v in out.items()})
