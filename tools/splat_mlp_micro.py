"""Split-attention micro-net kernels (fc1 -> bn1 -> relu -> fc2 on (B, C) vectors) on the model's configurations: us per call of
octa_splat_mlp_fwd / octa_splat_mlp_bwd.  Usage: python tools/splat_mlp_micro.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from architectures.models.octa import OctaScribbleNet
from octave_amd._lib import lib
from octave_amd.functional import _p, _st

B, H = 16, 400
dev = torch.device("cuda:0")
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
cfgs = collections.Counter()
for m in net.modules():
    if type(m).__name__ == "SplAtConv2d":
        inter, cg = m.fc1.weight.shape[0], m.fc1.weight.shape[1]
        card = m.cardinality
        cfgs[(cg * card, inter, card)] += 1
L = lib()


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tf = tb = 0.0
for (C, inter, card), n in sorted(cfgs.items()):
    f = lambda *s: torch.randn(*s, device=dev)
    gap, w1, b1 = f(B, C), f(inter, C // card) * 0.05, f(inter)
    g1, be1, rm, rv = f(inter), f(inter), torch.zeros(inter, device=dev), torch.ones(inter, device=dev)
    w2, b2 = f(2 * C, inter // card) * 0.05, f(2 * C)
    h1, h2 = torch.empty(B, inter, device=dev), torch.empty(B, inter, device=dev)
    mean, invstd, logits = torch.empty(inter, device=dev), torch.empty(inter, device=dev), torch.empty(B, 2 * C, device=dev)
    st = _st()
    fwd = lambda: L.octa_splat_mlp_fwd(_p(gap), _p(w1), _p(b1), _p(g1), _p(be1), _p(rm), _p(rv), 0.1, 1e-5, 1, _p(w2), _p(b2), _p(h1), _p(h2),
                                       _p(mean), _p(invstd), _p(logits), B, C, inter, card, st)
    dl, dh1, dgap = f(B, 2 * C), torch.empty(B, inter, device=dev), torch.zeros(B, C, device=dev)
    dw1, db1, dg, dbe, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(g1), torch.zeros_like(be1), torch.zeros_like(w2), torch.zeros_like(b2)
    bwd = lambda: L.octa_splat_mlp_bwd(_p(dl), _p(gap), _p(w1), _p(w2), _p(h1), _p(h2), _p(mean), _p(invstd), _p(g1), _p(dh1), _p(dgap), _p(dw1),
                                       _p(db1), _p(dg), _p(dbe), _p(dw2), _p(db2), B, C, inter, card, 0, st)
    a, b = timeit(fwd), timeit(bwd)
    tf += a * n; tb += b * n
    print(f"C {C:5d} inter {inter:4d} card {card} x{n}: fwd {a:6.1f} us (2 kernels)  bwd {b:6.1f} us (2-3 kernels)")
print(f"per step: fwd {tf / 1e3:.2f} ms, bwd {tb / 1e3:.2f} ms")
