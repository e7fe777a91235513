"""Split-attention micro-net backward (octa_splat_mlp_bwd: two launches) at the step's sizes: 30 back-to-back calls between two events.
Usage (GPU box): python tools/splat_mlp_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib

L = lib()
dev = torch.device("cuda:0")
p = F_._p
B = 16
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
SIZES = [(64, 32, 1), (128, 64, 1), (256, 128, 1), (512, 256, 1), (64, 32, 2), (128, 64, 2), (256, 128, 2), (512, 256, 2), (1024, 512, 2)]
for C, inter, card in SIZES:
    g = torch.Generator(device="cpu").manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    dlogits, gap = r(B, 2 * C), r(B, C)
    w1, w2 = r(inter, C // card), r(2 * C, inter // card)
    h1 = r(B, inter); h2 = torch.relu(r(B, inter))
    mean, invstd, gamma = r(inter), r(inter).abs() + 0.5, r(inter)
    dh1 = torch.empty(B, inter, device=dev); dgap = torch.zeros(B, C, device=dev)
    dw1, db1, dg, dbe, dw2, db2 = torch.zeros_like(w1), torch.zeros(inter, device=dev), torch.zeros(inter, device=dev), torch.zeros(inter, device=dev), torch.zeros_like(w2), torch.zeros(2 * C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: L.octa_splat_mlp_bwd(p(dlogits), p(gap), p(w1), p(w2), p(h1), p(h2), p(mean), p(invstd), p(gamma), p(dh1), p(dgap), p(dw1), p(db1), p(dg), p(dbe), p(dw2), p(db2),
                                      B, C, inter, card, 0, st)
    b1, beta, rm, rv = r(inter), r(inter), torch.zeros(inter, device=dev), torch.ones(inter, device=dev)
    b2, logits = r(2 * C), torch.empty(B, 2 * C, device=dev)
    ffn = lambda: L.octa_splat_mlp_fwd(p(gap), p(w1), p(b1), p(gamma), p(beta), p(rm), p(rv), 0.1, 1e-5, 1, p(w2), p(b2), p(h1), p(h2), p(mean), p(invstd), p(logits),
                                       B, C, inter, card, st)
    fcold = []
    for _ in range(7):
        evict.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ffn(); e1.record(); e1.synchronize()
        fcold.append(e0.elapsed_time(e1) * 1e3)
    fcold.sort()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ffn()
    e1.record(); e1.synchronize()
    fwarm = e0.elapsed_time(e1) * 1e3 / 30
    h2.clamp_(min=0)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
    cold = []
    for _ in range(7):
        evict.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    cold.sort()
    print(f"C {C:5d} inter {inter:4d} card {card}: {best:6.1f} us per call back to back, {cold[3]:6.1f} us behind a 512 MB sweep (AB + zero fill + CD); forward (2 launches) behind the sweep {fcold[3]:6.1f} us (all 7: {' '.join(f'{v:.0f}' for v in fcold)}), back to back {fwarm:5.1f} us; W2 {2 * C * inter // card * 4 / 1e6:.2f} MB, W1 {inter * C // card * 4 / 1e6:.2f} MB", flush=True)
