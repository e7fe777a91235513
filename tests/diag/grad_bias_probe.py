"""Where does the common factor of the HIP path's gradient norms at 48 x 48 come from?  (round-4 advisor finding: HIP fp32 sits
-1.8 .. -2.4 % from the float64 step on every parameter upstream of the attention maps, the reference's own fp32 run -0.63 %.)

Three measurements, HIP fp32 in DETERMINISTIC mode, oracle (= the reference's op sequence, pinned by its fixtures) on the CPU:

  (A) seeds: the same step on four different inputs.  A property of the KERNELS (a biased logf / expf, a lost term) keeps its sign and
      size on every input; one draw of amplified rounding noise per (implementation, input) does not.
  (B) the stage BEHIND the attention maps on identical inputs: the oracle's float64 (p, att) rounded to fp32 go into the HIP losses +
      discriminator as leaves; d(0.1 KL + 0.1 g_adv) / d(att_i), d / d(p) against the oracle's float64 gradients at the same point,
      per tensor: relative L2 error and norm ratio; the same restricted to the pixels whose attention probability is tiny (Q < 1e-4).
  (C) the network's backward on a FIXED cotangent: HIP forward, then backward of sum(att_i * c_i) + sum(agg * c) with the oracle's
      float64 cotangents c rounded to fp32; parameter-gradient norms against the float64 oracle run with the same cotangents.  What
      is left here is the segmentor's own forward + backward (conv, BatchNorm, split attention), no loss kernel involved.

Usage (GPU box): python tests/diag/grad_bias_probe.py > profiles/r05_grad_bias_probe.txt"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import ref_ops as R                                # noqa: E402
from oracle.fill import fill_state_dict, hash_input          # noqa: E402

Bn, H = 6, 48
UP = ("encoder", "decoder", "upsampling", "aag")             # parameters upstream of the attention maps


def inputs(seed):
    x = hash_input((Bn, 1, H, H), 1234 + seed).repeat(1, 3, 1, 1)
    u = hash_input((Bn, 1, H, H), 4321 + seed)
    ys = torch.zeros(Bn, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    return x, ys


def net_cpu():
    from architectures.models.octa import OctaScribbleNet
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
    fill_state_dict(net.state_dict())
    return net


def oracle_state(net, dt):
    P = {}
    for k, v in net.state_dict().items():
        v = v.detach().cpu().clone()
        if v.is_floating_point():
            v = v.to(dt)
            if not k.endswith(("running_mean", "running_var", "_u", "_v")):
                v.requires_grad_(True)
        P[k] = v
    return P


def draws(seed):
    """The CPU-generator draws one discriminator call consumes (discriminator/blocks.py:149-154, 165-170): noise plane, then the flip."""
    torch.manual_seed(seed)
    noise = torch.normal(mean=0.0, std=0.2, size=(H, H))
    flip = bool(torch.FloatTensor(1).uniform_(0, 1) < 0.1)
    return noise, flip


def oracle_step(net, x, ys, noise, flip, dt):
    P = oracle_state(net, dt)
    l, att, agg = R.segmentor_loss(P, x.to(dt), ys.to(dt), noise=noise.to(dt), flip=flip)
    l.backward()
    return {k[len("segmentor."):]: v.grad.double().norm().item() for k, v in P.items()
            if k.startswith("segmentor.") and v.requires_grad and v.grad is not None}, [a.detach() for a in att], agg.detach()


def hip_step(dev, x, ys, seed):
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    net = net_cpu().to(dev).train()
    att, agg, _ = net.segmentor(x.to(dev))
    torch.manual_seed(seed)            # the discriminator call below draws the same plane / flip from the CPU generator
    p = torch.softmax(agg, dim=1)
    ysd = ys.to(dev)
    kl = InterlayerDivergence()
    kl.check_nan = False
    l = net.supervised_loss(p, ysd) + DiceLoss()(p, ysd) + 0.1 * kl([p, *att]) + 0.1 * net.generator_loss(net.discriminator(att))
    net.zero_grad()
    l.backward()
    return {k: q.grad.double().norm().item() for k, q in net.segmentor.named_parameters() if q.grad is not None}


def common(gn, g64):
    top = max(g64.values())
    d = np.array([(gn[k] - g64[k]) / g64[k] for k in g64 if g64[k] > 1e-6 * top and k in gn and k.split("_")[0].split(".")[0].startswith(UP)])
    head = np.array([(gn[k] - g64[k]) / g64[k] for k in ("fc.weight", "fc.bias")])
    c = float(np.median(d))
    return c, float(np.median(np.abs(d - c))), float(np.abs(d - c).max()), float(np.abs(head).max())


def part_a(dev):
    print("(A) common factor of the gradient norms upstream of the attention maps, (implementation - float64) / float64, per input")
    print("    seed   HIP fp32: common  scatter(med/max)  head      oracle fp32: common  scatter(med/max)  head")
    for seed in (0, 1, 2, 3):
        x, ys = inputs(seed)
        noise, flip = draws(2024 + seed)
        net = net_cpu()
        g64, _, _ = oracle_step(net, x, ys, noise, flip, torch.float64)
        g32, _, _ = oracle_step(net, x, ys, noise, flip, torch.float32)
        gh = hip_step(dev, x, ys, 2024 + seed)
        ch, cr = common(gh, g64), common(g32, g64)
        print(f"    {seed}      {ch[0]:+.3e}  {ch[1]:.1e} / {ch[2]:.1e}  {ch[3]:.1e}      {cr[0]:+.3e}  {cr[1]:.1e} / {cr[2]:.1e}  {cr[3]:.1e}")


def part_b(dev):
    from architectures.segmentor.losses import InterlayerDivergence
    print("(B) the stage behind the attention maps on IDENTICAL inputs (the oracle's float64 p / att rounded to fp32): d(0.1 KL + 0.1 g_adv)")
    x, ys = inputs(0)
    noise, flip = draws(2024)
    net = net_cpu()
    P = oracle_state(net, torch.float64)
    with torch.no_grad():
        att, agg, _ = R.resnest_unet_forward(x.double(), {k: v.detach() for k, v in P.items()})
    p64 = F.softmax(agg, dim=1)
    leaves64 = [t.float().double().requires_grad_(True) for t in (p64, *att)]       # the fp32-rounded values, in float64
    l64 = 0.1 * R.interlayer_divergence(leaves64) + 0.1 * R.ls_generator_loss(R.discriminator_forward(leaves64[1:], {k: v.detach() for k, v in P.items()}, noise=noise.double(), flip=flip))
    l64.backward()
    hnet = net_cpu().to(dev).train()
    leaves = [t.detach().float().to(dev).requires_grad_(True) for t in leaves64]
    kl = InterlayerDivergence()
    kl.check_nan = False
    torch.manual_seed(2024)
    lh = 0.1 * kl(leaves) + 0.1 * hnet.generator_loss(hnet.discriminator(leaves[1:]))
    lh.backward()
    print(f"    loss: HIP {lh.item():.8f}  float64 {l64.item():.8f}")
    for i, (a, b) in enumerate(zip(leaves, leaves64)):
        gh, g6 = a.grad.double().cpu(), b.grad
        tiny = (b.detach() < 1e-4)
        rel = ((gh - g6).norm() / g6.norm()).item()
        rt = ((gh - g6)[tiny].norm() / max(g6[tiny].norm().item(), 1e-300)).item() if tiny.any() else float("nan")
        print(f"    {'p   ' if i == 0 else f'att{i - 1}'}: |g| ratio HIP / f64 {gh.norm().item() / g6.norm().item():.8f}  rel L2 err {rel:.2e}  "
              f"on the {int(tiny.sum())} tiny-Q entries: rel L2 err {rt:.2e}, their share of |g|^2 {float((g6[tiny] ** 2).sum() / (g6 ** 2).sum()):.3f}")


def part_c(dev):
    print("(C) the segmentor's forward + backward on FIXED cotangents (no loss kernel): parameter-gradient norms vs float64")
    x, _ = inputs(0)
    net = net_cpu()
    res = {}
    cot = None
    for dt in (torch.float64, torch.float32):
        P = oracle_state(net, dt)
        att, agg, _ = R.resnest_unet_forward(x.to(dt), P)
        outs = [agg, *att]
        if cot is None:
            cot = [hash_input(tuple(o.shape), 7300 + i, -1.0, 1.0) for i, o in enumerate(outs)]
        sum((o * c.to(dt)).sum() for o, c in zip(outs, cot)).backward()
        res[dt] = {k[len("segmentor."):]: v.grad.double().norm().item() for k, v in P.items()
                   if k.startswith("segmentor.") and v.requires_grad and v.grad is not None}
    hnet = net_cpu().to(dev).train()
    att, agg, _ = hnet.segmentor(x.to(dev))
    sum((o * c.to(dev)).sum() for o, c in zip([agg, *att], cot)).backward()
    gh = {k: q.grad.double().norm().item() for k, q in hnet.segmentor.named_parameters() if q.grad is not None}
    ch, cr = common(gh, res[torch.float64]), common(res[torch.float32], res[torch.float64])
    print(f"    HIP fp32: common {ch[0]:+.3e} scatter med {ch[1]:.1e} max {ch[2]:.1e} head {ch[3]:.1e};  oracle fp32: common {cr[0]:+.3e} scatter med {cr[1]:.1e} max {cr[2]:.1e} head {cr[3]:.1e}")


if __name__ == "__main__":
    from octave_amd import functional as F_
    dev = torch.device("cuda:0")
    F_.set_deterministic(True)
    torch.set_num_threads(16)
    part_a(dev)
    part_b(dev)
    part_c(dev)
