"""Backward-pass reproducibility per OP at encoder_3 geometry (B = 16, 25 x 25): one forward, the same cotangent fed to backward
twice (retain_graph), max |diff| / max |value| per output.  Localises which op turns float-atomic rounding noise into percent-level
differences.  Usage: python tools/repro_ops.py [f32|bf16]"""
import sys

import torch

sys.path.insert(0, ".")
from architectures.segmentor.compose import ResnestUNet       # noqa: E402
from octave_amd import functional as F_                       # noqa: E402


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def twice(fn, inputs, tag):
    """fn(*inputs) -> tensor; grads wrt inputs twice with the same cotangent"""
    ins = [i.detach().clone().requires_grad_(True) for i in inputs]
    y = fn(*ins)
    y = y[0] if isinstance(y, tuple) else y
    g = torch.randn(tuple(y.shape), generator=torch.Generator(device="cpu").manual_seed(9)).to(y.device).to(y.dtype)
    g = F_.to_nhwc(g) if g.dim() == 4 else g
    g1 = torch.autograd.grad(y, ins, g, retain_graph=True, allow_unused=True)
    g2 = torch.autograd.grad(y, ins, g, retain_graph=True, allow_unused=True)
    torch.cuda.synchronize()
    out = [f"{rel(a, b):.2e}" for a, b in zip(g1, g2) if a is not None]
    print(f"   {tag:34s} grads of inputs (two backward calls, one forward): {out}", flush=True)


def main():
    dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    unet = ResnestUNet(2, False).to(dev).train()
    blk = unet.encoder_3[1]                                   # identity-shortcut bottleneck, 1024 channels at 25 x 25
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = F_.to_nhwc(torch.randn(16, 1024, 25, 25, generator=gen).to(dev).to(dt))
    print(f"== {dt}")
    twice(lambda a: blk(a), [x], "whole bottleneck")
    twice(lambda a: blk.conv1(a), [x], "conv1 1x1 1024->256 (dgrad)")
    h = blk.conv1(x).detach()
    twice(lambda a: blk.bn1(a, relu=True), [h], "bn1 + relu")
    h = blk.bn1(h, relu=True).detach()
    sp = blk.conv2
    twice(lambda a: sp.conv(a), [h], "splat grouped 3x3 (dgrad)")
    h2 = sp.conv(h).detach()
    twice(lambda a: sp.bn0(a, relu=True), [h2], "splat bn0 + relu")
    h2 = sp.bn0(h2, relu=True).detach()

    def tail(a):
        return F_.splat_tail(a, sp.fc1.weight, sp.fc1.bias, sp.bn1.weight, sp.bn1.bias, sp.bn1.running_mean, sp.bn1.running_var, sp.fc2.weight, sp.fc2.bias,
                             sp.cardinality, 0.1, 1e-5, True, False)
    twice(tail, [h2], "split-attention tail")
    # the tail with a cotangent that is identical for every sample and pixel (the attention branch's share of dx is then pure cancellation)
    a = h2.detach().clone().requires_grad_(True)
    y = tail(a)
    for name, g in (("random cotangent", None), ("constant cotangent", torch.ones_like(y))):
        if g is None:
            g = F_.to_nhwc(torch.randn(tuple(y.shape), generator=torch.Generator(device="cpu").manual_seed(9)).to(dev).to(y.dtype))
        d1 = torch.autograd.grad(y, [a], g, retain_graph=True)[0]
        d2 = torch.autograd.grad(y, [a], g, retain_graph=True)[0]
        print(f"   tail, {name:20s} dx rel diff {rel(d1, d2):.2e}  |dx| max {float(d1.float().abs().max()):.3e} mean {float(d1.float().abs().mean()):.3e}")
    with torch.no_grad():
        gap = F_.to_nchw_f32(h2).view(16, 2, -1, 625).sum(1).mean(-1)          # (B, C)
        print(f"   gap across the batch: mean |value| {float(gap.abs().mean()):.3e}, std over batch (mean over channels) {float(gap.std(0).mean()):.3e}")


if __name__ == "__main__":
    main()
