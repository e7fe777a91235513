"""GPU parity tests, module level: the drop-in ``architectures`` modules (HIP path) against the golden
vectors produced by the reference (tests/golden) and against the oracle, through the same public
nn.Module API a user of the reference would call."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BAND, BAND_GRAD

from oracle.fill import fill_state_dict, hash_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need the MI355X (run with -m gpu on the GPU box)")
    return torch.device("cuda:0")


def check(name, got, want, rtol, atol):
    got = got.detach().float().cpu()
    want = torch.as_tensor(np.asarray(want)).float() if not isinstance(want, torch.Tensor) else want.detach().float().cpu()
    assert tuple(got.shape) == tuple(want.shape), f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, got.shape) if got.dim() else ()
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} out of tolerance; worst at {idx}: got {got[idx].item():.6g} "
                             f"want {want[idx].item():.6g} (|err| {err[idx].item():.3g}, max|want| {want.abs().max().item():.3g})")


def run_block(G, tag, mod, x, dev, rel=1e-4, gtol=10.0):
    """Outputs within rel * max|reference| (north_star: 1e-4 fp32); gradients within gtol times that
    (the 3-sample BatchNorm inside split attention amplifies rounding noise in the backward)."""
    fill_state_dict(mod.state_dict())
    mod = mod.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    out = mod(xd)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    loss = 0
    for i, o in enumerate(outs):
        check(f"{tag} out{i}", o, G[f"{tag}/out{i}"], 0, rel * float(np.abs(G[f"{tag}/out{i}"]).max()))
        loss = loss + (o.float() * hash_input(tuple(o.shape), 7000 + i, -1, 1).to(dev)).sum()
    loss.backward()
    check(f"{tag} grad_x", xd.grad, G[f"{tag}/grad_x"], 0, rel * gtol * float(np.abs(G[f"{tag}/grad_x"]).max()))
    params = dict(mod.named_parameters())
    bufs = dict(mod.named_buffers())
    n = 0
    for k, g in G.items():
        if k.startswith(f"{tag}/grad/"):
            name = k[len(f"{tag}/grad/"):]
            assert params[name].grad is not None, f"{tag}: no grad for {name}"
            if name.endswith(("fc1.bias", "conv2.conv.bias", "conv.3.conv.bias")) or name == "conv.bias":
                # a bias directly in front of a BatchNorm has an analytically ZERO gradient: the reference's
                # value is rounding noise (|g| ~ 1e-6), so only its smallness is checked
                assert float(params[name].grad.abs().max()) < 1e-3 * max(1.0, float(np.abs(G[f"{tag}/grad_x"]).max())), name
                n += 1
                continue
            check(k, params[name].grad, g, 0, rel * gtol * float(np.abs(g).max()) + 1e-6)
            n += 1
        if k.startswith(f"{tag}/buf/") and not k.endswith("num_batches_tracked"):
            check(k, bufs[k[len(f"{tag}/buf/"):]], g, 1e-4, 1e-5)
        if k.endswith("num_batches_tracked") and k.startswith(f"{tag}/buf/"):
            assert int(bufs[k[len(f"{tag}/buf/"):]]) == int(g), k
    assert n > 0


def test_blocks_vs_reference_golden(dev, golden):
    from torch import nn
    from architectures.extra.resnest import Bottleneck, ResNestDecoder, SplAtConv2d, Upsampling, _AvgDown, _ConvBN
    from architectures.segmentor.blocks import AdversarialAttentionGate
    from octave_amd.layers import BatchNorm2d, Conv2d
    G = golden("blocks.npz")
    run_block(G, "splat_enc", SplAtConv2d(16, 16, 3, padding=1, groups=1, bias=False, radix=2, norm_layer=BatchNorm2d),
              hash_input((3, 16, 6, 5), 11, -1, 1), dev)
    run_block(G, "splat_dec", SplAtConv2d(32, 32, 3, padding=1, stride=1, groups=2, radix=2, norm_layer=BatchNorm2d),
              hash_input((3, 32, 5, 7), 12, -1, 1), dev)
    down = _ConvBN(_AvgDown(2), Conv2d(32, 64, 1, bias=False), BatchNorm2d(64))
    run_block(G, "bottleneck_s2", Bottleneck(32, 16, stride=2, downsample=down, radix=2, cardinality=1, bottleneck_width=64, avd=True,
                                             avd_first=False, norm_layer=BatchNorm2d), hash_input((3, 32, 8, 8), 13, -1, 1), dev)
    run_block(G, "bottleneck_id", Bottleneck(64, 16, radix=2, cardinality=1, bottleneck_width=64, avd=True, avd_first=False,
                                             norm_layer=BatchNorm2d), hash_input((3, 64, 5, 5), 14, -1, 1), dev)
    run_block(G, "decoder", ResNestDecoder(64, 32), hash_input((3, 64, 6, 6), 15, -1, 1), dev)
    run_block(G, "upsampling", Upsampling(16, 8), hash_input((2, 16, 5, 3), 16, -1, 1), dev)
    run_block(G, "aag", AdversarialAttentionGate(32, 2), hash_input((2, 32, 7, 5), 17, -1, 1), dev)
    run_block(G, "aag3", AdversarialAttentionGate(16, 3), hash_input((2, 16, 4, 4), 18, -1, 1), dev)


# ----------------------------------------------------------------------------- losses vs reference golden
B, C, H, W = 3, 2, 16, 16


def probs(seed, shape=(B, C, H, W)):
    return F.softmax(3.0 * hash_input(shape, seed, -1, 1), dim=1)


def scribble(seed, shape=(B, C, H, W), empty_class=None):
    u = hash_input((shape[0], 1, shape[2], shape[3]), seed)
    ys = torch.zeros(shape)
    ys[:, 1:2] = (u < 0.08).float()
    ys[:, 0:1] = ((u > 0.3) & (u < 0.4)).float()
    if empty_class is not None:
        ys[:, empty_class] = 0
    return ys


@pytest.mark.parametrize("case", ["all", "stopgrad", "zero_weight"])
def test_interlayer_kl_backward_adds_parked_fanout_gradients(dev, case):
    """octa_interlayer_kl_bwd_add (round 5): every attention map feeds InterlayerDivergence AND a second consumer (the discriminator in the
    training step, models/octa.py); the second consumer's gradient is parked by stash_grad and added where the KL gradient is written.
    Against autograd's own sums without holders.  `stopgrad`: no basis gradient from the KL term -- the parked one must still arrive;
    `zero_weight`: a map the divergence skips (weight 0) still gets its parked gradient."""
    from octave_amd import functional as F_
    weights = [1, 0.5, 0, 2, 1] if case == "zero_weight" else [1] * 5
    other_w = [hash_input((B, C, 32 >> max(i - 1, 0), 32 >> max(i - 1, 0)), 70 + i, -1, 1).to(dev) for i in range(6)]

    def run(fused):
        att = [probs(60 + i, (B, C, 32 >> max(i - 1, 0), 32 >> max(i - 1, 0))).to(dev).requires_grad_(True) for i in range(6)]
        hs = [F_.GradHolder() for _ in att] if fused else None
        kl = F_.interlayer_kl(att, weights, case == "stopgrad", holders=hs)[0]
        second = [F_.stash_grad(a, h) for a, h in zip(att, hs)] if fused else att          # created AFTER the divergence's node
        other = sum((a * a * w).sum() for a, w in zip(second, other_w))
        (kl + 0.3 * other).backward()
        if fused:
            assert all(h.consumed and h.grad is None for h in hs)
        return [a.grad.clone() for a in att]
    got, want = run(True), run(False)
    for i, (g, w) in enumerate(zip(got, want)):
        check(f"kl fan-out {case} grad{i}", g, w, 1e-6, 1e-7 * float(w.abs().max()) + 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(64, 64, 1, 4, 24, 20, False), (128, 64, 2, 3, 33, 17, True), (32, 256, 1, 16, 13, 13, False)])
def test_splat_with_bn0_on_the_fly_vs_separate_batchnorm(dev, dtype, cfg):
    """SplAtConv2d in training mode recomputes bn0 + ReLU inside the split-attention kernels (octa_splat_bn_*: no BatchNorm-apply
    pass, no stored activation); OCTA_FUSE_SPLAT_BN0=0 runs conv -> BatchNorm -> split attention as separate ops.  Same module, same
    input and cotangent, both ways, two steps: outputs, input gradient, every parameter gradient and every buffer.  fp32: the two
    differ by summation order only; 16-bit: the separate path rounds the BatchNorm output to the storage type, the fused one does not
    (bound: a few ulp of the activation scale on outputs, 2 % of the gradient scale on gradients)."""
    import architectures.extra.resnest as R
    from octave_amd import functional as F_
    from octave_amd.layers import BatchNorm2d
    cin, ch, card, B, H, W, relu_after = cfg
    gen = torch.Generator(device="cpu").manual_seed(31)
    x0 = (torch.randn(B, cin, H, W, generator=gen) * 1.5 + 0.3)
    g0 = torch.randn(B, ch, H, W, generator=gen)

    def run(fused):
        old = R._FUSE_SPLAT_BN0
        R._FUSE_SPLAT_BN0 = fused
        try:
            torch.manual_seed(5)
            m = R.SplAtConv2d(cin, ch, 3, padding=1, groups=card, bias=True, radix=2, norm_layer=BatchNorm2d).to(dev).train()
            with torch.no_grad():
                m.bn0.weight.copy_(torch.rand(2 * ch, generator=torch.Generator().manual_seed(7)) + 0.5)
                m.bn0.bias.copy_(torch.randn(2 * ch, generator=torch.Generator().manual_seed(8)) * 0.3)
            outs = []
            for it in range(2):
                for p_ in m.parameters():
                    p_.grad = None
                x = F_.to_nhwc(x0.to(dev), dtype=dtype).detach().requires_grad_(True)
                y = m(x, relu_after)
                (y.float() * g0.to(dev)).sum().backward()
                outs.append((F_.to_nchw_f32(y.detach()), F_.to_nchw_f32(x.grad), {n: p_.grad.detach().float().clone() for n, p_ in m.named_parameters()},
                             {n: b_.detach().float().clone() for n, b_ in m.named_buffers()}))
            return outs
        finally:
            R._FUSE_SPLAT_BN0 = old
    a, b = run(True), run(False)
    f32 = dtype == torch.float32
    for it in range(2):
        ya, dxa, ga, ba = a[it]
        yb, dxb, gb, bb = b[it]
        sy, sdx = float(yb.abs().max()), float(dxb.abs().max())
        assert float((ya - yb).abs().max()) <= (2e-5 if f32 else 2.5e-2) * sy, (it, float((ya - yb).abs().max()), sy)
        rel = float((dxa - dxb).norm() / dxb.norm())
        assert rel <= (2e-4 if f32 else 2e-2), (it, "dx", rel)
        gmax = max(float(v.abs().max()) for v in gb.values())
        for n in gb:
            if n in ("conv.bias", "fc1.bias"):
                # a bias in front of a BatchNorm (bn0 resp. bn1): the exact gradient is 0, both paths return rounding noise
                assert float(ga[n].abs().max()) <= (1e-3 if f32 else 5e-2) * gmax and float(gb[n].abs().max()) <= (1e-3 if f32 else 5e-2) * gmax, (it, n)
                continue
            rel = float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-3 * gmax * gb[n].numel() ** 0.5))
            assert rel <= (1e-3 if f32 else 3e-2), (it, n, rel)
        for n in bb:
            if n.endswith("num_batches_tracked"):
                assert int(ba[n]) == int(bb[n]) == it + 1, (it, n)
            else:
                assert float((ba[n] - bb[n]).abs().max()) <= (1e-5 if f32 else 2e-3) * (float(bb[n].abs().max()) + 1.0), (it, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(64, 64, 1, 4, 24, 20, False), (128, 64, 2, 3, 33, 17, True), (32, 256, 1, 16, 13, 13, False)])
def test_splat_softmax_backward_inside_the_micro_net_backward(dev, dtype, cfg):
    """octa_splat_bn_bwd_da2 + octa_splat_mlp_bwd_da (round 5: the radix-2 softmax backward of resnest.py:126-127 applied where the micro-net's
    backward reads the attention gradients, no launch of its own) against the separate splat_softmax_bwd_kernel (OCTA_SPLAT_SOFTMAX_INLINE=0),
    deterministic mode: the same fp32 formula on the same sums -- input gradient and every parameter gradient within 1e-5 of the tensor's scale
    (an fma contracted differently is all that may differ)."""
    import architectures.extra.resnest as R
    from octave_amd import functional as F_
    from octave_amd.layers import BatchNorm2d
    cin, ch, card, B, H, W, relu_after = cfg
    gen = torch.Generator(device="cpu").manual_seed(37)
    x0 = (torch.randn(B, cin, H, W, generator=gen) * 1.5 + 0.3)
    g0 = torch.randn(B, ch, H, W, generator=gen)

    def run(inline):
        old = F_._SPLAT_SOFTMAX_INLINE
        F_._SPLAT_SOFTMAX_INLINE = inline
        try:
            torch.manual_seed(5)
            m = R.SplAtConv2d(cin, ch, 3, padding=1, groups=card, bias=True, radix=2, norm_layer=BatchNorm2d).to(dev).train()
            x = F_.to_nhwc(x0.to(dev), dtype=dtype).detach().requires_grad_(True)
            y = m(x, relu_after)
            (y.float() * g0.to(dev)).sum().backward()
            return F_.to_nchw_f32(y.detach()), F_.to_nchw_f32(x.grad), {n: p_.grad.detach().float().clone() for n, p_ in m.named_parameters()}
        finally:
            F_._SPLAT_SOFTMAX_INLINE = old
    F_.set_deterministic(True)
    try:
        (ya, dxa, ga), (yb, dxb, gb) = run(True), run(False)
    finally:
        F_.set_deterministic(False)
    assert torch.equal(ya, yb)
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # (16 bit: dx is rounded to the storage type after the fp32 arithmetic)
    assert float((dxa - dxb).abs().max()) <= tol * float(dxb.abs().max()), float((dxa - dxb).abs().max())
    gmax = max(float(v.abs().max()) for v in gb.values())
    for n in gb:
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-5 * max(float(gb[n].abs().max()), 1e-3 * gmax) + 1e-9, (n, float((ga[n] - gb[n]).abs().max()), float(gb[n].abs().max()))


def test_losses_vs_reference_golden(dev, golden):
    from architectures.discriminator.losses import LSDiscriminatorialLoss, LSGeneratorLoss
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence, WeightedPartialCE
    G = golden("losses.npz")
    wpce = WeightedPartialCE(num_classes=2, manual=True)
    for tag, kw in [("wpce_mean", {}), ("wpce_sum", {"reduction": "sum"}), ("wpce_full", {"full": True}),
                    ("wpce_ignore_bg", {"ignore_bg": True})]:
        p = probs(21).to(dev).requires_grad_(True)
        ys = scribble(22).to(dev)
        l = wpce(p, ys, **kw)
        l.backward()
        check(f"{tag} loss", l, G[f"{tag}/loss"], 2e-5, 1e-6)
        check(f"{tag} grad", p.grad, G[f"{tag}/grad"], 2e-4, 1e-7)
        check(f"{tag} ys_after", ys, G[f"{tag}/ys_after"], 0, 0)
    p = probs(21).to(dev).requires_grad_(True)
    l = wpce(p, scribble(22, empty_class=1).to(dev))
    l.backward()
    check("wpce_empty loss", l, G["wpce_empty/loss"], 2e-5, 1e-6)
    check("wpce_empty grad", p.grad, G["wpce_empty/grad"], 2e-4, 1e-7)
    with pytest.raises(AssertionError):
        wpce(torch.rand(1, 2, 4, 4, device=dev), torch.rand(1, 3, 4, 4, device=dev))

    for tag, ps, t in [("dice", 23, scribble(24)),
                       ("dice_dense", 25, F.one_hot((hash_input((B, H, W), 26) > 0.7).long(), 2).permute(0, 3, 1, 2).float())]:
        p = probs(ps).to(dev).requires_grad_(True)
        l = DiceLoss()(p, t.to(dev))
        l.backward()
        check(f"{tag} loss", l, G[f"{tag}/loss"], 1e-5, 1e-7)
        check(f"{tag} grad", p.grad, G[f"{tag}/grad"], 2e-4, 1e-9)

    def pyramid(seed0, H0=32, n=6):
        return [probs(seed0 + i, (B, C, H0 >> max(i - 1, 0), H0 >> max(i - 1, 0))).to(dev).requires_grad_(True) for i in range(n)]
    for tag, kw, w in [("kl_default", {}, None), ("kl_stopgrad", {"stop_gradient": True}, None), ("kl_weights", {}, [1, 0.5, 0, 2, 1]),
                       ("kl_short_weights", {}, [1, 2, 3, 4, 5, 6, 7])]:
        att = pyramid(30)
        l = InterlayerDivergence(**kw)(att, w)
        l.backward()
        check(f"{tag} loss", l, G[f"{tag}/loss"], 5e-5, 1e-6)
        for i, a in enumerate(att):
            g = a.grad if a.grad is not None else torch.zeros_like(a)
            check(f"{tag} grad{i}", g, G[f"{tag}/grad{i}"], 5e-4, 1e-8)
    with pytest.raises(NotImplementedError):
        InterlayerDivergence(mode="sum")(pyramid(30))
    bad = pyramid(30)
    bad[2] = torch.full_like(bad[2], float("nan"))
    with pytest.raises(Exception, match="NaN"):
        InterlayerDivergence()(bad)

    r = hash_input((4, 1), 41, -2, 2).to(dev).requires_grad_(True)
    f = hash_input((4, 1), 42, -2, 2).to(dev).requires_grad_(True)
    l = LSDiscriminatorialLoss()(r, f)
    l.backward()
    check("lsd loss", l, G["lsd/loss"], 1e-5, 1e-7)
    check("lsd grad_real", r.grad, G["lsd/grad_real"], 1e-5, 1e-7)
    check("lsd grad_fake", f.grad, G["lsd/grad_fake"], 1e-5, 1e-7)
    f2 = hash_input((4, 1), 43, -2, 2).to(dev).requires_grad_(True)
    l = LSGeneratorLoss()(f2)
    l.backward()
    check("lsg loss", l, G["lsg/loss"], 1e-5, 1e-7)
    check("lsg grad", f2.grad, G["lsg/grad"], 1e-5, 1e-7)


# ----------------------------------------------------------------------------- discriminator vs reference golden
def test_discriminator_vs_reference_golden(dev, golden):
    from architectures.discriminator.blocks import DiscriminatorBlock
    G = golden("disc.npz")
    Bd, Hd = 2, 64
    m = DiscriminatorBlock(torch.Size((Bd, 2, Hd, Hd)), is_training=True, depth=4, num_filters=8)
    fill_state_dict(m.state_dict())
    m = m.to(dev).train()

    def pyr(seed):
        return [F.softmax(2 * hash_input((Bd, 2, Hd >> i, Hd >> i), seed + i, -1, 1), dim=1).to(dev).requires_grad_(True) for i in range(5)]
    for call in range(2):
        torch.manual_seed(100 + call)              # the forward must consume normal(H,W) then uniform(1) from the CPU generator
        ys = pyr(50 + 10 * call)
        out = m(ys)
        m.zero_grad()
        (out * hash_input(tuple(out.shape), 60 + call, -1, 1).to(dev)).sum().backward()
        check(f"disc call{call} out", out, G[f"call{call}/out"], 5e-4, 1e-5)
        for i, y in enumerate(ys):
            check(f"disc call{call} grad_y{i}", y.grad, G[f"call{call}/grad_y{i}"], 2e-3, 1e-6)
        params, bufs = dict(m.named_parameters()), dict(m.named_buffers())
        for k, g in G.items():
            if k.startswith(f"call{call}/grad/"):
                check(k, params[k.split("/grad/")[1]].grad, g, 2e-3, 1e-5)
            if k.startswith(f"call{call}/buf/"):
                check(k, bufs[k.split("/buf/")[1]], g, 1e-4, 1e-6)
    m.eval()
    torch.manual_seed(7)
    with torch.no_grad():
        out = m([y.detach() for y in pyr(90)])
    check("disc eval out", out, G["eval/out"], 5e-4, 1e-5)


# ----------------------------------------------------------------------------- whole network vs reference golden
def _scribble(Bn, Hn):
    u = hash_input((Bn, 1, Hn, Hn), 4321)
    ys = torch.zeros(Bn, 2, Hn, Hn)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    return ys


def _build(Bn, Hn, dev):
    from architectures.models.octa import OctaScribbleNet
    net = OctaScribbleNet(torch.Size((Bn, 3, Hn, Hn)), torch.Size((Bn, 2, Hn, Hn)), True, False)
    fill_state_dict(net.state_dict())
    P = {k: v.clone() for k, v in net.state_dict().items()}
    return net.to(dev).train(), P


@pytest.mark.parametrize("Hn", [48, 64])
def test_unet_stagewise_vs_oracle_fp32(dev, Hn):
    """north_star parity bar, stage by stage: every stage of the oracle is fed the HIP path's own input
    to that stage and the outputs must agree within 1e-4 * max|oracle| (fp32).  Feeding each stage the
    same input keeps the comparison free of the chaotic amplification discussed in DESIGN.md."""
    from oracle import ref_ops as R
    Bn = 3
    net, P = _build(Bn, Hn, dev)
    seg = net.segmentor
    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap[name] = ([i.detach().float().cpu() for i in inp if torch.is_tensor(i)],
                         [o.detach().float().cpu() for o in (out if isinstance(out, tuple) else (out,))])
        return f
    names = ["encoder_0_2_2", "encoder_1", "encoder_2", "encoder_3", "encoder_4"] + \
            [f"{p}_{i}" for i in range(5) for p in ("upsampling", "decoder", "aag")]
    for n in names:
        getattr(seg, n).register_forward_hook(hook(n))
    x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1)
    att, agg, x4 = seg(x.to(dev))
    Q = lambda: {k: v.clone() for k, v in P.items()}   # noqa: E731  (fresh buffers: BN updates them in place)

    def cmp(name, got, want, rel=1e-4):
        check(f"{name} [{Hn}]", got, want, 0, rel * float(want.abs().max()) + 1e-7)
    with torch.no_grad():
        cmp("stem", cap["encoder_0_2_2"][0][0], R.stem(x, Q(), "segmentor.encoder_0_1_2"))
        cmp("maxpool", cap["encoder_0_2_2"][1][0], F.max_pool2d(cap["encoder_0_2_2"][0][0], 3, 2, 1), rel=0)
        for i in range(4):
            n = f"encoder_{i + 1}"
            cmp(n, cap[n][1][0], R.encoder_stage(cap[n][0][0], Q(), "segmentor." + n, i))
        for i in range(5):
            n = f"upsampling_{i}"
            cmp(n, cap[n][1][0], R.upsampling(cap[n][0][0], Q(), "segmentor." + n))
            n = f"decoder_{i}"
            cmp(n, cap[n][1][0], R.resnest_decoder(cap[n][0][0], Q(), "segmentor." + n))
            n = f"aag_{i}"
            m, y = R.attention_gate(cap[n][0][0], Q(), "segmentor." + n)
            cmp(n + " masked", cap[n][1][0], m)
            cmp(n + " y_hat", cap[n][1][1], y)
        d0 = cap["aag_0"][1][0]
        cmp("head fc", agg, F.conv2d(d0, P["segmentor.fc.weight"], P["segmentor.fc.bias"]))
    assert tuple(x4.shape) == (Bn, 2048, (Hn // 16 + 1) // 2, (Hn // 16 + 1) // 2)
    assert [tuple(a.shape[2:]) for a in att] == [(Hn >> i, Hn >> i) for i in range(5)]


STAGES = [
    # name, input shape, oracle function
    ("encoder_1", (8, 64, 12, 12), lambda R, x, P, n: R.encoder_stage(x, P, n, 0)),
    ("encoder_2", (8, 256, 12, 12), lambda R, x, P, n: R.encoder_stage(x, P, n, 1)),
    ("encoder_4", (8, 1024, 4, 4), lambda R, x, P, n: R.encoder_stage(x, P, n, 3)),
    ("decoder_4", (4, 2048, 3, 3), lambda R, x, P, n: R.resnest_decoder(x, P, n)),
    ("decoder_2", (4, 512, 12, 12), lambda R, x, P, n: R.resnest_decoder(x, P, n)),
    ("decoder_0", (2, 64, 32, 32), lambda R, x, P, n: R.resnest_decoder(x, P, n)),
    ("upsampling_4", (4, 2048, 2, 2), lambda R, x, P, n: R.upsampling(x, P, n)),
    ("upsampling_0", (2, 64, 16, 16), lambda R, x, P, n: R.upsampling(x, P, n)),
    ("aag_4", (4, 1024, 3, 3), lambda R, x, P, n: R.attention_gate(x, P, n)),
    ("aag_0", (2, 32, 32, 32), lambda R, x, P, n: R.attention_gate(x, P, n)),
]


@pytest.mark.parametrize("stage", STAGES, ids=[s[0] for s in STAGES])
def test_real_width_stage_fwd_bwd_vs_oracle(dev, stage):
    """Forward AND backward of each U-Net stage at its real channel widths (tile edges of the MFMA
    kernels: N up to 2048, K up to 18432) against the oracle on the same seeded input/cotangent."""
    from oracle import ref_ops as R
    name, shape, fn = stage
    net, P = _build(2, 48, dev)
    mod = getattr(net.segmentor, name)
    pref = "segmentor." + name
    Ps = {k: v.clone() for k, v in P.items() if k.startswith(pref + ".")}
    for k, v in Ps.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x = hash_input(shape, 77, -1, 1)
    xr = x.clone().requires_grad_(True)
    outs_r = fn(R, xr, Ps, pref)
    outs_r = outs_r if isinstance(outs_r, tuple) else (outs_r,)
    cots = [hash_input(tuple(o.shape), 88 + i, -1, 1) for i, o in enumerate(outs_r)]
    sum((o * c).sum() for o, c in zip(outs_r, cots)).backward()
    # the same stage in fp64: the anchor that says how far plain fp32 itself sits from the exact gradient
    P64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in Ps.items()}
    for k, v in P64.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    outs_64 = fn(R, x64, P64, pref)
    outs_64 = outs_64 if isinstance(outs_64, tuple) else (outs_64,)
    sum((o * c.double()).sum() for o, c in zip(outs_64, cots)).backward()
    xd = x.to(dev).requires_grad_(True)
    outs = mod(xd)
    outs = outs if isinstance(outs, tuple) else (outs,)
    sum((o.float() * c.to(dev)).sum() for o, c in zip(outs, cots)).backward()
    for i, (o, w) in enumerate(zip(outs, outs_r)):
        check(f"{name} out{i}", o, w, 0, 1e-4 * float(w.abs().max()))

    def grad_check(what, got, want):
        """Gradients of a ReLU network are discontinuous: where a pre-activation sits within rounding distance
        of zero the two implementations may pick different masks, which changes that pixel's gradient by O(1) and
        (through the BatchNorm sums) its whole channel by ~1e-4.  Verified on encoder_2/B=8: the HIP data-gradient
        equals torch's conv_transpose2d on the same incoming gradient to 1e-7 while both differ from the oracle's
        end-to-end value at one pixel row; that perturbation then spreads at the 1e-3 level through the earlier
        blocks' BatchNorms.  So: median error <= 3e-3 of the scale, 99.5 % of the elements within 1e-2, all within 0.1."""
        g, w = got.detach().float().cpu(), want.detach().float()
        scale = float(w.abs().max()) + 1e-12
        err = (g - w).abs() / scale
        assert torch.isfinite(g).all(), what
        nbad = int((err > 1e-2).sum())
        frac = nbad / err.numel()
        assert float(err.median()) <= 3e-3 and nbad <= max(2, 5e-3 * err.numel()) and float(err.max()) <= 0.1, \
            f"{what}: median {float(err.median()):.2g}, {frac:.2%} of elements off by > 1e-2, max {float(err.max()):.3g} (relative to max|grad|)"
    def anchored(what, got, ref32, ref64):
        """fp64-anchored bound on the whole gradient tensor (the forward test's band, in the L2 norm so that a ReLU mask that
        flips in one implementation and not in the other is weighed, not vetoed): the HIP gradient may be at most 4x as far
        from the fp64 gradient as the fp32 oracle is, plus 1e-3 of the gradient's norm for the fp32 summation order."""
        g, r32, r64 = got.detach().double().cpu(), ref32.detach().double(), ref64.detach().double()
        nrm = float(r64.norm()) + 1e-30
        e_hip, e_32 = float((g - r64).norm()) / nrm, float((r32 - r64).norm()) / nrm
        assert e_hip <= 4.0 * e_32 + 1e-3, f"{what}: |hip - fp64| = {e_hip:.3g} of |grad|, fp32 oracle sits at {e_32:.3g}"
    grad_check(f"{name} grad_x", xd.grad, xr.grad)
    anchored(f"{name} grad_x", xd.grad, xr.grad, x64.grad)
    for k, pm in mod.named_parameters():
        want = Ps[pref + "." + k].grad
        if k.endswith(("fc1.bias", "conv2.conv.bias", "conv.3.conv.bias")):
            continue        # analytically zero gradient (bias in front of a BatchNorm)
        assert pm.grad is not None and want is not None, k
        grad_check(f"{name} grad {k}", pm.grad, want)
        anchored(f"{name} grad {k}", pm.grad, want, P64[pref + "." + k].grad)
    for k, b in mod.named_buffers():
        if not k.endswith("num_batches_tracked"):
            check(f"{name} buffer {k}", b, Ps[pref + "." + k], 1e-4, 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_side_branch_shortcuts_match_the_main_stream_path(dev, dtype):
    """The opt-in second-stream shortcuts (OCTA_SIDE_SHORTCUT / OCTA_SIDE_SHORTCUT_DEC: a bottleneck's avg-down shortcut and a decoder
    block's 1x1 shortcut issued beside the main branch, functional.SideBranch) against the default single-stream order, network level,
    forward + backward with the deferred weight-gradient queue on, in DETERMINISTIC mode: logits bit-identical, every gradient bit-identical
    in fp32 and within 2e-2 (relative L2) in bf16.  (Round
    4's advisor: nothing under tests/ enabled the switch; side-branch convs now launch their weight gradients on their own stream instead
    of parking x / dy -- allocated in the side stream's pool -- for a flush on the main stream.)"""
    from architectures.extra import resnest as RN
    from architectures.segmentor.losses import DiceLoss
    from octave_amd import functional as F_
    Bn, Hn = 3, 64
    x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1).to(dev)
    ys = _scribble(Bn, Hn).to(dev)

    def run(side):
        old = (RN._SIDE_SHORTCUT, RN._SIDE_SHORTCUT_DEC, F_._FUSE_FANOUT_SKIP)
        RN._SIDE_SHORTCUT = RN._SIDE_SHORTCUT_DEC = side
        # (a pool on the side stream does not take the skip connection's fan-out addend -- functional.take_fanout -- so the comparison
        # runs both orders with autograd's own add there: the bit-identity below is about the stream order, not about that fusion)
        F_._FUSE_FANOUT_SKIP = False
        F_.defer_wgrads(True)
        try:
            net, _ = _build(Bn, Hn, dev)
            net.segmentor.compute_dtype = dtype
            att, agg, _ = net.segmentor(x)
            p = torch.softmax(agg.float(), dim=1)
            (net.supervised_loss(p, ys) + DiceLoss()(p, ys) + sum(a.float().square().mean() for a in att)).backward()
            F_.flush_wgrads()
            torch.cuda.synchronize()
            return agg.detach().clone(), {k: q.grad.detach().clone() for k, q in net.segmentor.named_parameters() if q.grad is not None}
        finally:
            F_.defer_wgrads(False)
            RN._SIDE_SHORTCUT, RN._SIDE_SHORTCUT_DEC, F_._FUSE_FANOUT_SKIP = old

    F_.set_deterministic(True)
    try:
        a0, g0 = run(False)
        a1, g1 = run(True)
    finally:
        F_.set_deterministic(False)
    assert torch.isfinite(a1).all() and set(g0) == set(g1)
    assert torch.equal(a0, a1), float((a0.float() - a1.float()).abs().max())           # same forward kernels in both orders
    if dtype == torch.float32:
        bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
        assert not bad, f"{len(bad)} gradients differ with the side-stream shortcuts: {bad[:6]}"
    else:
        # 16-bit: a side-branch conv's weight gradient runs on the single-problem kernel instead of the batched 8-wave one (another
        # summation order), everything else is identical
        for k in g0:
            d, n = (g0[k].double() - g1[k].double()).norm().item(), g0[k].double().norm().item()
            assert d <= 2e-2 * n + 1e-6, (k, d, n)


@pytest.mark.parametrize("Hn", [48, 64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_skip_connection_fusions_match_the_separate_ops(dev, dtype, Hn):
    """Round 5's two skip-connection fusions, network level, against their parity twins (OCTA_FUSE_UPCAT=0 / OCTA_FUSE_FANOUT_SKIP=0):
    the transposed conv storing into its slice of the cat buffer (same kernel, another output stride: logits bit-identical), and the
    pool backward kernels adding the cat's gradient slice (one rounding fewer than autograd's separate add in 16 bit; the same fp32
    sums in another association in fp32).  48 x 48 takes the cropped cat at the bottom of the U (x_3 is 3 x 3: padded to 4 x 4 for
    encoder_4, whose up-sampled output is cropped back to 3 x 3), 64 x 64 the uncropped one everywhere."""
    from architectures.segmentor.losses import DiceLoss
    from octave_amd import functional as F_
    Bn = 3
    x = hash_input((Bn, 1, Hn, Hn), 4321).repeat(1, 3, 1, 1).to(dev)
    ys = _scribble(Bn, Hn).to(dev)

    def run(fused):
        old = (F_._FUSE_UPCAT, F_._FUSE_FANOUT_SKIP)
        F_._FUSE_UPCAT = F_._FUSE_FANOUT_SKIP = fused
        F_.defer_wgrads(True)
        try:
            net, _ = _build(Bn, Hn, dev)
            net.segmentor.compute_dtype = dtype
            att, agg, _ = net.segmentor(x)
            p = torch.softmax(agg.float(), dim=1)
            (net.supervised_loss(p, ys) + DiceLoss()(p, ys) + sum(a.float().square().mean() for a in att)).backward()
            F_.flush_wgrads()
            torch.cuda.synchronize()
            return agg.detach().clone(), {k: q.grad.detach().clone() for k, q in net.segmentor.named_parameters() if q.grad is not None}
        finally:
            F_.defer_wgrads(False)
            F_._FUSE_UPCAT, F_._FUSE_FANOUT_SKIP = old

    F_.set_deterministic(True)
    try:
        a0, g0 = run(False)
        a1, g1 = run(True)
    finally:
        F_.set_deterministic(False)
    assert torch.isfinite(a1).all() and set(g0) == set(g1)
    assert torch.equal(a0, a1), float((a0.float() - a1.float()).abs().max())
    # decoder-side gradients never see the fused sums: bit-identical; the encoder's differ by the association / rounding of one add
    # (train-mode BatchNorm over 3 samples amplifies the changed rounding on its way down the encoder: 1e-3 of a tensor's norm in fp32)
    # 16 bit: one bf16 rounding per skip tensor element differs, and the 3-sample BatchNorms make that a per-tensor deviation of a few
    # per cent with a tail (measured: 6 % on the stem's BatchNorm bias at 48 x 48) -- the logic is pinned by the fp32 run; here the
    # median over the tensors and a loose per-tensor cap
    tol = 2e-3 if dtype == torch.float32 else 0.5
    gmax = max(float(g.abs().max()) for g in g0.values())
    rel = sorted((g0[k].double() - g1[k].double()).norm().item() / (g0[k].double().norm().item() + 1e-12) for k in g0
                 if not k.endswith(("fc1.bias", "conv2.conv.bias", "conv.3.conv.bias")))
    assert rel[len(rel) // 2] <= (1e-4 if dtype == torch.float32 else 3e-2), rel[len(rel) // 2]
    for k in g0:
        d, n = (g0[k].double() - g1[k].double()).norm().item(), g0[k].double().norm().item()
        if k.startswith(("decoder_", "upsampling_", "aag_", "fc.")):
            assert torch.equal(g0[k], g1[k]), (k, d, n)
        elif k.endswith(("fc1.bias", "conv2.conv.bias", "conv.3.conv.bias")):
            assert float(g1[k].abs().max()) < 1e-3 * max(1.0, gmax), k      # a bias in front of a BatchNorm: analytically zero, noise
        else:
            assert d <= tol * n + 1e-7, (k, d, n)


@pytest.mark.parametrize("Hn", [48, 64])
def test_unet_vs_reference_golden_fp32(dev, golden, Hn):
    """End to end against the reference's CPU fp32 result.  93 train-mode BatchNorms (some over 3
    samples) amplify fp32 rounding noise: the reference's OWN fp32 output is max|ref32-ref64| away from
    its float64 evaluation (stored in the fixture).  The HIP fp32 path must (a) be as close to the
    float64 reference as the fp32 reference itself is (factor BAND = 2.5, conftest.py), and (b) produce the same argmax mask
    wherever the decision margin exceeds that noise."""
    from architectures.segmentor.losses import DiceLoss
    G = golden(f"unet_{Hn}.npz")
    Bn = 3
    net, _ = _build(Bn, Hn, dev)
    x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1).to(dev)
    att, agg, x4 = net.segmentor(x)
    noise = float(np.abs(G["agg"] - G["agg_f64"]).max())
    scale = float(np.abs(G["agg"]).max())
    e64 = float(np.abs(agg.detach().cpu().numpy().astype(np.float64) - G["agg_f64"]).max())
    e32 = float(np.abs(agg.detach().cpu().numpy() - G["agg"]).max())
    print(f"[unet {Hn}] logits: |hip-ref64| {e64:.3e}  |ref32-ref64| {noise:.3e}  ratio {e64 / noise:.2f}  |hip-ref32| {e32:.3e}  scale {scale:.1f}")
    assert e64 <= BAND * noise + 1e-4 * scale, (e64, noise)
    assert e32 <= (BAND + 1) * noise + 1e-4 * scale, (e32, noise)
    for i, a in enumerate(att):
        n_i = float(np.abs(G[f"att{i}"] - G[f"att{i}_f64"]).max())
        check(f"att{i}", a, G[f"att{i}_f64"], 0, BAND * n_i + 1e-4)
    want_arg = np.argmax(G["agg"], axis=1)
    got_arg = torch.argmax(agg, dim=1).cpu().numpy()
    margin = np.abs(G["agg"][:, 0] - G["agg"][:, 1])
    safe = margin > 10 * noise
    assert np.array_equal(got_arg[safe], want_arg[safe]), "argmax mask differs outside the rounding-noise band"
    print(f"[unet {Hn}] argmax: {(got_arg != want_arg).sum()} of {got_arg.size} pixels differ (all inside the noise band)")
    if Hn == 48:
        onehot = net.segmentor.predict(x, "one-hot")[1].cpu().numpy().astype(np.uint8)
        assert np.array_equal(onehot[:, 1][safe], G["onehot"][:, 1][safe])
        net, _ = _build(Bn, Hn, dev)     # predict() ran a second train-mode forward: rebuild for the backward check
        att, agg, x4 = net.segmentor(x)
    ys = _scribble(Bn, Hn).to(dev)
    p = torch.softmax(agg, dim=1)
    loss = net.supervised_loss(p, ys) + DiceLoss()(p, ys)
    loss.backward()
    l64, l32 = float(G["loss_f64"]), float(G["loss"])
    assert abs(loss.item() - l64) <= BAND * abs(l32 - l64) + 1e-4 * abs(l64), (loss.item(), l32, l64)
    params, bufs = dict(net.segmentor.named_parameters()), dict(net.segmentor.named_buffers())
    norms = {}
    for k, g in G.items():
        if k.startswith("gradnorm_f64/"):
            name = k[len("gradnorm_f64/"):]
            assert params[name].grad is not None, name
            g64 = float(g)
            if g64 < 1e-9:          # analytically zero (a bias in front of a BatchNorm): rounding noise only
                continue
            norms[name] = (params[name].grad.double().norm().item(), float(G["gradnorm/" + name]), g64)
        if k.startswith("buf/"):
            check(k, bufs[k[4:]], g, 2e-3, 1e-4)
    # the backward is as chaotic as the forward: the reference's own fp32 gradient norms sit up to
    # worst_ref away from its float64 evaluation; the HIP path must stay within 4x that band.
    # Deviations are relative to max(|g64|, 1e-4 * largest norm): a gradient that is 1e-5 of the others (aag_3.conv1.bias at
    # 64^2: 2.4e-5 against 4.6, a sum of cancelling terms) is pure noise in BOTH implementations -- the reference's own fp32
    # value is 18 % off, and the HIP value moves between 0.3 and 1.4 from run to run with the order of the float atomics in
    # the weight gradients (tools/grad_flake.py) -- so its RELATIVE deviation says nothing; the floor turns it into an
    # absolute bound at the scale of the gradients that matter.
    floor = 1e-4 * max(n[2] for n in norms.values())
    devs = {k: (abs(h - g64) / max(g64, floor), abs(r - g64) / max(g64, floor)) for k, (h, r, g64) in norms.items()}
    worst_hip = max(d[0] for d in devs.values())
    worst_ref = max(d[1] for d in devs.values())
    med_hip = float(np.median([d[0] for d in devs.values()]))
    med_ref = float(np.median([d[1] for d in devs.values()]))
    assert worst_hip <= BAND_GRAD * worst_ref + 2e-3, (worst_hip, worst_ref)
    assert med_hip <= BAND_GRAD * med_ref + 1e-3, (med_hip, med_ref)
    for k in G["nograd_keys"].tolist():
        assert params[k].grad is None, f"{k} must not receive a gradient"
    print(f"[unet {Hn}] grad norms vs float64 reference: worst/median deviation HIP {worst_hip:.2e}/{med_hip:.2e}, reference fp32 {worst_ref:.2e}/{med_ref:.2e}")


def test_unet_bf16_sane(dev, golden):
    """bf16 activations: finite, same argmax as the fp32 reference on all but low-margin pixels."""
    from architectures.models.octa import OctaScribbleNet
    G = golden("unet_64.npz")
    Bn, Hn = 3, 64
    net = OctaScribbleNet(torch.Size((Bn, 3, Hn, Hn)), torch.Size((Bn, 2, Hn, Hn)), True, False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    net.segmentor.compute_dtype = torch.bfloat16
    x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1).to(dev)
    att, agg, x4 = net.segmentor(x)
    assert torch.isfinite(agg).all()
    err = (agg.detach().cpu().numpy() - G["agg"])
    print(f"[bf16] logits max abs err {np.abs(err).max():.3e}, rms {np.sqrt((err ** 2).mean()):.3e}, ref rms {np.sqrt((G['agg'] ** 2).mean()):.3e}")
    margin = np.abs(G["agg"][:, 0] - G["agg"][:, 1])
    got_arg = torch.argmax(agg, dim=1).cpu().numpy()
    want_arg = np.argmax(G["agg"], axis=1)
    big = margin > 0.25 * np.abs(G["agg"]).max()
    assert (got_arg[big] == want_arg[big]).mean() > 0.98
    ys = _scribble(Bn, Hn).to(dev)
    out = net.supervised_loss(torch.softmax(agg, 1), ys)
    out.backward()
    for k, p in net.segmentor.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k


def test_product_path_has_no_cpu_fallback():
    from octave_amd import functional as F_
    from octave_amd._lib import OctaError
    with pytest.raises(OctaError):
        F_.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 1, 1))
