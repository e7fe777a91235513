"""GPU parity tests, kernel level: every C-ABI op of libocta_hip.so against plain torch CPU ops
(the primitives the oracle is written in) on seeded inputs, fp32 (tight) and bf16 (loose)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need the MI355X (run with -m gpu on the GPU box)")
    return torch.device("cuda:0")


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(shape, generator=g)


def check(name, got, want, rtol, atol):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, got.shape)
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} out of tolerance; worst at {idx}: got {got[idx].item():.6g} "
                             f"want {want[idx].item():.6g} (|err| {err[idx].item():.3g}, max|want| {want.abs().max().item():.3g})")


TOL = {torch.float32: dict(rtol=2e-4, atol=2e-5), torch.bfloat16: dict(rtol=3e-2, atol=3e-2), torch.float16: dict(rtol=4e-3, atol=4e-3)}


# --------------------------------------------------------------------------- layout probes
def test_mfma_and_tr16_layouts(dev):
    from octave_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    a = torch.randint(-4, 5, (16, 32)).float()
    b = torch.randint(-4, 5, (32, 16)).float()
    d = torch.zeros(16, 16, device=dev)
    ad, bd = a.to(dev).bfloat16(), b.to(dev).bfloat16()      # keep references: data_ptr() of a temporary dangles
    L.octa_probe_mfma(0, ad.data_ptr(), bd.data_ptr(), d.data_ptr(), st)
    check("mfma bf16 16x16x32 layout", d, a @ b, 0, 0)
    a = torch.randint(-4, 5, (16, 4)).float()
    b = torch.randint(-4, 5, (4, 16)).float()
    ad, bd = a.to(dev), b.to(dev)
    L.octa_probe_mfma(1, ad.data_ptr(), bd.data_ptr(), d.data_ptr(), st)
    check("mfma f32 16x16x4 layout", d, a @ b, 0, 0)
    p = torch.randint(-4, 5, (32, 16)).float()
    q = torch.randint(-4, 5, (32, 16)).float()
    pd, qd = p.to(dev).bfloat16(), q.to(dev).bfloat16()
    L.octa_probe_mfma(2, pd.data_ptr(), qd.data_ptr(), d.data_ptr(), st)
    check("ds_read_tr16_b64 transposed operand", d, p.t() @ q, 0, 0)


# --------------------------------------------------------------------------- conv engine
CONV_CASES = [
    # Cin, Cout, k, s, p, g, B, H, W
    (16, 32, 1, 1, 0, 1, 3, 7, 5),
    (32, 64, 3, 1, 1, 1, 2, 9, 6),
    (32, 64, 3, 1, 1, 2, 2, 6, 9),
    (64, 128, 3, 1, 1, 4, 2, 5, 5),
    (32, 64, 3, 1, 1, 4, 2, 8, 8),       # decoder_0 SplAt conv: 8 in / 16 out per group
    (3, 32, 3, 2, 1, 1, 2, 16, 16),      # stem (Cin padded 3 -> 8)
    (2, 8, 4, 2, 1, 1, 2, 16, 16),       # discriminator stack_0
    (15, 16, 4, 2, 1, 1, 2, 8, 8),       # spectral conv (Cin padded 15 -> 16)
    (64, 13, 1, 1, 0, 1, 2, 6, 6),       # squeeze conv (Cout 13)
    (256, 200, 1, 1, 0, 1, 1, 13, 11),   # N-tile edge
    (128, 64, 3, 1, 1, 1, 2, 20, 20),    # long K, several M tiles
    (24, 40, 3, 2, 1, 1, 2, 11, 9),      # odd sizes, stride 2
    (64, 16, 3, 1, 1, 1, 2, 9, 19),      # 3x3 halo kernel: N<=16, partial tiles
    (64, 48, 3, 1, 1, 1, 2, 17, 33),     # 3x3 halo kernel: N<=64, partial tiles in both axes
    (96, 256, 3, 1, 1, 1, 1, 8, 16),     # 3x3 halo kernel: two N tiles, 3 channel chunks
    (128, 256, 3, 1, 1, 2, 2, 13, 13),   # 3x3 halo kernel, grouped (encoder SplAt shape)
    (128, 512, 1, 1, 0, 1, 8, 6, 6),     # strided-bottleneck conv3 at M = 288 (64x64 tiles, 5 x 8 grid)
    (256, 128, 1, 1, 0, 1, 8, 12, 12),   # strided-bottleneck conv1
    (256, 512, 1, 1, 0, 1, 8, 6, 6),     # shortcut 1x1 after the 2x2 average pool
    (128, 256, 3, 1, 1, 2, 8, 12, 12),   # its split-attention conv
    (32, 64, 3, 1, 1, 4, 1, 130, 140),   # decoder_0 SplAt conv at high resolution: densified (block-diagonal) halo path
    (64, 128, 3, 1, 1, 4, 1, 130, 140),  # decoder_1 SplAt conv at high resolution: pairs of groups merged (two groups of 32 -> 64), halo wgrad with channel sets
    (128, 128, 3, 1, 1, 2, 1, 128, 144), # grouped halo weight gradient, one channel set per group (64 -> 64 per group)
    (64, 32, 3, 1, 1, 1, 1, 130, 140),   # halo weight-gradient kernel, N <= 32 (two channel chunks per block), partial tiles
    (128, 64, 3, 1, 1, 1, 1, 128, 136),  # halo weight-gradient kernel, N <= 64, four channel-chunk blocks
    (32, 48, 3, 1, 1, 1, 2, 128, 128),   # halo weight-gradient kernel, N = 48 (partial n-tile), batch 2
    (32, 32, 3, 1, 1, 1, 1, 136, 128),   # halo weight-gradient kernel, stem shape (N = 32 with one channel chunk)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, dtype, case):
    from octave_amd import functional as F_
    Cin, Cout, k, s, p, g, B, H, W = case
    x = rnd((B, Cin, H, W), 1)
    w = rnd((Cout, Cin // g, k, k), 2) * (1.0 / (k * k * Cin / g) ** 0.5)
    b = rnd((Cout,), 3)
    if dtype != torch.float32:   # compare against the same rounded operands
        x, w = x.to(dtype).float(), w.to(dtype).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p, groups=g)
    gy = rnd(tuple(yr.shape), 4)
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    (yr * gy).sum().backward()

    xd = x.to(dev).to(dtype).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = F_.conv2d(xd, wd, bd, s, p, g)
    assert y.shape == yr.shape
    t = TOL[dtype]
    check(f"conv fwd {case}", y, yr, **t)
    (y.float() * gy.to(dev)).sum().backward()
    check(f"conv dgrad {case}", xd.grad, xr.grad, **t)
    sc = float(B * yr.shape[2] * yr.shape[3]) ** 0.5
    check(f"conv wgrad {case}", wd.grad, wr.grad, rtol=t["rtol"], atol=t["atol"] * sc)
    check(f"conv bias grad {case}", bd.grad, br.grad, rtol=t["rtol"], atol=t["atol"] * sc)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv2d_channels_last_weight_and_act(dev, dtype):
    from octave_amd import functional as F_
    from octave_amd._lib import ACT_TANH
    x = rnd((2, 16, 6, 6), 5)
    w = rnd((24, 16, 3, 3), 6) * 0.1
    if dtype != torch.float32:
        x, w = x.to(dtype).float(), w.to(dtype).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = torch.tanh(F.conv2d(xr, wr, None, padding=1))
    gy = rnd(tuple(yr.shape), 7)
    (yr * gy).sum().backward()
    xd = x.to(dev).to(dtype).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F_.conv2d(xd, wd, None, 1, 1, 1, ACT_TANH)
    (y.float() * gy.to(dev)).sum().backward()
    t = TOL[dtype]
    check("conv+tanh fwd", y, yr, **t)
    check("conv+tanh dgrad", xd.grad, xr.grad, **t)
    check("conv+tanh wgrad (channels-last dw)", wd.grad, wr.grad, rtol=t["rtol"], atol=t["atol"] * 10)
    assert wd.grad.stride() == wd.stride()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(16, 8, 2, 5, 3), (64, 32, 2, 6, 6), (24, 40, 1, 4, 7)])
def test_conv_transpose2x2(dev, dtype, shape):
    from octave_amd import functional as F_
    Cin, Cout, B, H, W = shape
    x = rnd((B, Cin, H, W), 8)
    w = rnd((Cin, Cout, 2, 2), 9) * 0.2
    b = rnd((Cout,), 10)
    if dtype != torch.float32:
        x, w = x.to(dtype).float(), w.to(dtype).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    gy = rnd(tuple(yr.shape), 11)
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    (yr * gy).sum().backward()
    xd = x.to(dev).to(dtype).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = F_.conv_transpose2x2(xd, wd, bd)
    (y.float() * gy.to(dev)).sum().backward()
    t = TOL[dtype]
    check("convT fwd", y, yr, **t)
    check("convT dgrad", xd.grad, xr.grad, **t)
    check("convT wgrad", wd.grad, wr.grad, rtol=t["rtol"], atol=t["atol"] * 10)
    check("convT bias grad", bd.grad, br.grad, rtol=t["rtol"], atol=t["atol"] * 10)


# --------------------------------------------------------------------------- batch norm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(32, 3, 7, 5, True, True), (64, 2, 6, 6, False, False), (2048, 2, 2, 2, True, False),
                                 (40, 4, 1, 1, True, False), (512, 2, 9, 9, False, True)])
def test_batch_norm(dev, dtype, cfg):
    from octave_amd import functional as F_
    C, B, H, W, relu, use_res = cfg
    x = rnd((B, C, H, W), 12, -2, 3)
    res = rnd((B, C, H, W), 13)
    gam, bet = rnd((C,), 14, 0.5, 1.5), rnd((C,), 15)
    rm, rv = rnd((C,), 16), rnd((C,), 17, 0.5, 1.5)
    if dtype != torch.float32:
        x, res = x.to(dtype).float(), res.to(dtype).float()
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if use_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    gy = rnd(tuple(yr.shape), 18)
    (yr * gy).sum().backward()
    xd = x.to(dev).to(dtype).requires_grad_(True)
    rd = res.to(dev).to(dtype).requires_grad_(True)
    gd, bd = gam.to(dev).requires_grad_(True), bet.to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    y = F_.batch_norm(xd, gd, bd, rmd, rvd, 0.1, 1e-5, True, relu, rd if use_res else None)
    (y.float() * gy.to(dev)).sum().backward()
    t = TOL[dtype]
    check("bn fwd", y, yr, **t)
    check("bn running_mean", rmd, rmr, rtol=1e-4, atol=1e-5)
    check("bn running_var", rvd, rvr, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=1e-5)
    n = B * H * W
    check("bn dx", xd.grad, xr.grad, rtol=t["rtol"] * 5, atol=t["atol"] * 5)
    check("bn dgamma", gd.grad, gr.grad, rtol=t["rtol"] * 5, atol=t["atol"] * n ** 0.5)
    check("bn dbeta", bd.grad, br.grad, rtol=t["rtol"] * 5, atol=t["atol"] * n ** 0.5)
    if use_res:
        check("bn dres", rd.grad, rr.grad, **t)


def test_batch_norm_single_value_raises(dev):
    from octave_amd import functional as F_
    x = torch.zeros(1, 8, 1, 1, device=dev)
    with pytest.raises(ValueError):
        F_.batch_norm(x, torch.ones(8, device=dev), torch.zeros(8, device=dev), None, None, 0.1, 1e-5, True)


# --------------------------------------------------------------------------- pooling / copies
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_pools(dev, dtype):
    from octave_amd import functional as F_
    t = TOL[dtype]
    for (B, C, H, W) in [(2, 16, 8, 8), (2, 8, 9, 7), (1, 64, 12, 12)]:
        x = rnd((B, C, H, W), 20, -1, 1)
        x[:, :, ::3, ::2] = 0.0   # ties, like post-ReLU maps
        if dtype != torch.float32:
            x = x.to(dtype).float()
        for name, fr, fg in [
            ("maxpool3s2", lambda a: F.max_pool2d(a, 3, 2, 1), lambda a: F_.max_pool3s2(a)),
            ("avgpool3s2p1", lambda a: F.avg_pool2d(a, 3, 2, 1), lambda a: F_.avg_pool(a, 3, 2, 1)),
            ("avgpool2s2ceil", lambda a: F.avg_pool2d(a, 2, 2, ceil_mode=True, count_include_pad=False),
             lambda a: F_.avg_pool(a, 2, 2, 0, True, False)),
        ]:
            xr = x.clone().requires_grad_(True)
            yr = fr(xr)
            gy = rnd(tuple(yr.shape), 21)
            (yr * gy).sum().backward()
            xd = x.to(dev).to(dtype).requires_grad_(True)
            y = fg(xd)
            (y.float() * gy.to(dev)).sum().backward()
            check(f"{name} fwd {B,C,H,W}", y, yr, **t)
            check(f"{name} bwd {B,C,H,W}", xd.grad, xr.grad, **t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_backward_adds_a_parked_fanout_gradient(dev, dtype):
    """octa_maxpool3s2_bwd_add / octa_avgpool_bwd_add (round 5): x feeds a pool AND a second consumer (a channel slice of a wider
    cat, the U-Net skip of compose.py:141-147); the second consumer's gradient is parked by stash_grad and added by the pool's
    backward kernel.  Against torch on CPU, and against autograd's own sum with the holder withheld."""
    from octave_amd import functional as F_
    t = TOL[dtype]
    for (B, C, H, W) in [(2, 16, 8, 8), (2, 8, 9, 7), (1, 64, 13, 13)]:
        x = rnd((B, C, H, W), 40, -1, 1)
        x[:, :, ::3, ::2] = 0.0
        other = rnd((B, 8, H, W), 41)
        if dtype != torch.float32:
            x, other = x.to(dtype).float(), other.to(dtype).float()
        for name, fr, fg in [
            ("maxpool3s2", lambda a: F.max_pool2d(a, 3, 2, 1), lambda a, h: F_.max_pool3s2(a, h)),
            ("avgpool2s2ceil", lambda a: F.avg_pool2d(a, 2, 2, ceil_mode=True, count_include_pad=False),
             lambda a, h: F_.avg_pool(a, 2, 2, 0, True, False, h)),
        ]:
            xr = x.clone().requires_grad_(True)
            yr, cr = fr(xr), torch.cat((xr, other), 1)
            gy, gc = rnd(tuple(yr.shape), 42), rnd(tuple(cr.shape), 43)
            ((yr * gy).sum() + (cr * gc).sum()).backward()
            grads = []
            for fused in (True, False):
                xd = F_.to_nhwc(x.to(dev), dtype=dtype).requires_grad_(True)
                od = F_.to_nhwc(other.to(dev), dtype=dtype)
                h = F_.offer_fanout(xd) if fused else None
                assert (h is not None) == fused
                if fused:
                    h = F_.take_fanout()
                y = fg(xd, h)
                c = F_.cat_crop(F_.skip_with_fanout(xd, h), od)        # the slice of c's gradient is a strided view
                assert (type(c.grad_fn.next_functions[0][0]).__name__ == "StashGradFnBackward") == fused
                ((y.float() * gy.to(dev)).sum() + (c.float() * gc.to(dev)).sum()).backward()
                assert h is None or (h.consumed and h.grad is None)
                check(f"{name}+fanout bwd {B,C,H,W} fused={fused}", xd.grad, xr.grad, **t)
                grads.append(xd.grad.float().cpu())
            if dtype == torch.float32:
                check(f"{name}+fanout fused vs autograd's add", grads[0], grads[1], 1e-6, 1e-6)


def test_cat_pad_crop(dev):
    from octave_amd import functional as F_
    a = rnd((2, 16, 4, 4), 22)
    b = rnd((2, 8, 4, 4), 23)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.cat((ar, br), 1)[:, :, :-1, :-1]
    gy = rnd(tuple(yr.shape), 24)
    (yr * gy).sum().backward()
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = F_.cat_crop(ad, bd, 3, 3)
    (y * gy.to(dev)).sum().backward()
    check("cat+crop", y, yr, 0, 0)
    check("cat+crop grad a", ad.grad, ar.grad, 0, 0)
    check("cat+crop grad b", bd.grad, br.grad, 0, 0)
    x = rnd((2, 8, 3, 3), 25)
    xr = x.clone().requires_grad_(True)
    yr = F.pad(xr, (0, 1, 0, 1))
    gy = rnd(tuple(yr.shape), 26)
    (yr * gy).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    y = F_.pad_bottom_right(xd, 1, 1)
    (y * gy.to(dev)).sum().backward()
    check("pad", y, yr, 0, 0)
    check("pad grad", xd.grad, xr.grad, 0, 0)


# --------------------------------------------------------------------------- loss kernels (vs oracle)
def test_losses_vs_oracle_random(dev):
    from oracle import ref_ops as R
    from octave_amd import functional as F_
    B, K, H, W = 3, 2, 40, 24
    logits = rnd((B, K, H, W), 30, -3, 3)
    u = rnd((B, 1, H, W), 31, 0, 1)
    ys = torch.zeros(B, K, H, W)
    ys[:, 1:2] = (u < 0.1).float()
    ys[:, 0:1] = ((u > 0.4) & (u < 0.5)).float()
    lr = logits.clone().requires_grad_(True)
    p = F.softmax(lr, 1)
    want = R.weighted_partial_ce(p, ys, K) + 0.7 * R.dice_loss(p, ys)
    want.backward()
    ld = logits.to(dev).requires_grad_(True)
    out = F_.wpce_dice(ld, ys.to(dev), from_logits=True)
    got = out[0] + 0.7 * out[1]
    got.backward()
    check("fused softmax+wpce+dice", got, want, 2e-5, 1e-6)
    check("fused loss grad", ld.grad, lr.grad, 2e-4, 1e-8)
    # non-contiguous (NHWC-strided) probabilities
    pd = F.softmax(logits, 1).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pr = F.softmax(logits, 1).requires_grad_(True)
    w2 = R.weighted_partial_ce(pr, ys, K)
    w2.backward()
    g2 = F_.wpce_dice(pd, ys.to(dev))[0]
    g2.backward()
    check("wpce strided", g2, w2, 2e-5, 1e-6)
    check("wpce strided grad", pd.grad, pr.grad, 2e-4, 1e-8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_repack_all_matches_per_weight_packs(dev, dtype):
    """octa_pack_many (one launch, linear + LDS-transposed tiles) must reproduce the per-weight pack kernels bit for bit:
    forward, data-gradient and conv-transpose operands, channels-last and plain OIHW parameters, grouped, padded, odd sizes;
    the dense block-diagonal kinds are checked against an explicitly densified weight."""
    from octave_amd import functional as F_
    g = torch.Generator().manual_seed(3)
    specs = [  # (Cout, Cin_g, k, groups, channels_last)
        (64, 32, 3, 1, True), (48, 24, 3, 1, False), (256, 64, 1, 1, True), (128, 64, 3, 2, True), (64, 8, 3, 4, True),
        (13, 64, 1, 1, True), (40, 3, 3, 1, True), (96, 15, 4, 1, True), (200, 136, 1, 1, True), (32, 70, 3, 1, True)]
    params, jobs = [], []
    for O, Ig, k, gr, cl in specs:
        w = torch.randn(O, Ig, k, k, generator=g).to(dev)
        if cl:
            w = w.contiguous(memory_format=torch.channels_last)
        p = torch.nn.Parameter(w)
        params.append(p)
        jobs += [(p, "fwd", gr, F_.round8(Ig)), (p, "dgrad", gr, F_.round8(O // gr))]
    wt = torch.nn.Parameter(torch.randn(72, 40, 2, 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last))
    params.append(wt)
    jobs.append((wt, "convT", 1, F_.round8(72)))
    wd = params[4]                                          # (64, 8, 3, 3), groups 4 -> dense 32 -> 64
    jobs += [(wd, "fwd_dense", 4, 32), (wd, "dgrad_dense", 4, 64)]
    first = [F_._packed(p, kind, dtype, gr, pad).clone() for p, kind, gr, pad in jobs]      # per-weight kernels
    with torch.no_grad():
        for p in params:
            p.mul_(1.5).add_(0.25)
    F_.bump_weight_epoch()
    n = F_.repack_all(params)
    assert n == sum(not F_._PACK_CACHE[(id(p), kind, dtype, gr, pad)].direct for p, kind, gr, pad in jobs)
    torch.cuda.synchronize()
    multi = [F_._PACK_CACHE[(id(p), kind, dtype, gr, pad)].out.clone() for p, kind, gr, pad in jobs]
    F_._PACK_CACHE.clear()
    F_._PACK_PLANS.clear()
    single = [F_._packed(p, kind, dtype, gr, pad) for p, kind, gr, pad in jobs]
    for (p, kind, gr, pad), a, b, f in zip(jobs, multi, single, first):
        if F_._PACK_CACHE[(id(p), kind, dtype, gr, pad)].direct:
            continue
        assert a.shape == b.shape and torch.equal(a, b), (kind, tuple(p.shape), gr, pad, (a.float() - b.float()).abs().max().item())
        assert not torch.equal(a, f)
    # dense kinds against an explicitly block-diagonal weight packed as an ordinary dense conv
    dense = torch.zeros(64, 32, 3, 3, device=dev)
    for gi in range(4):
        dense[gi * 16:(gi + 1) * 16, gi * 8:(gi + 1) * 8] = wd.detach()[gi * 16:(gi + 1) * 16]
    dense = dense.contiguous(memory_format=torch.channels_last)
    ref_fwd = F_._packed(dense, "fwd", dtype, 1, 32)
    if ref_fwd.dim() == 4:          # fp32: the channels-last parameter storage itself is the operand ([O][KH][KW][I] in memory)
        ref_fwd = ref_fwd.permute(0, 2, 3, 1)
    assert torch.equal(ref_fwd.reshape(-1), single[-2].reshape(-1))
    assert torch.equal(F_._packed(dense, "dgrad", dtype, 1, 64).reshape(-1), single[-1].reshape(-1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 32, 37, 29), (2, 13, 9, 7), (2, 264, 5, 6), (1, 2048, 3, 3), (4, 8, 50, 50)])
def test_colsum(dev, dtype, shape):
    """bias-gradient column sums (16-byte vector kernel when C % 8 == 0, scalar kernel otherwise); accumulates into `out`."""
    from octave_amd import functional as F_
    B, C, H, W = shape
    x = rnd(shape, 21)
    if dtype != torch.float32:
        x = x.to(dtype).float()
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    out = torch.full((C,), 0.5, dtype=torch.float32, device=dev)
    F_.raw_colsum(xd, out)
    ref = x.double().sum(dim=(0, 2, 3)).float() + 0.5
    check(f"colsum {shape}", out, ref, rtol=1e-4, atol=1e-4 * (B * H * W) ** 0.5)


# --------------------------------------------------------------------------- full-size properties (BASELINE sizes: B=16, 400x400)
FULL_SIZE_CONVS = [
    # Cin, Cout, k, s, p, g, B, H, W                     the kernels this shape goes through
    (64, 32, 3, 1, 1, 1, 16, 400, 400),    # decoder_0 3x3: halo fwd/dgrad 128x32 / 128x64, halo weight-gradient kernel <2>
    (32, 64, 3, 1, 1, 4, 16, 400, 400),    # decoder_0 split-attention conv: densified block-diagonal fwd/dgrad, halo wgrad <4> (diagonal)
    (64, 128, 3, 1, 1, 4, 16, 200, 200),   # decoder_1 split-attention conv: pair-merged grouped resident-weight fwd/dgrad, halo wgrad <4> over two channel sets
    (2, 64, 4, 2, 1, 1, 16, 400, 400),     # discriminator stack_0: generic 256x64, strided dgrad = GEMM + col2im
    (64, 32, 1, 1, 0, 1, 16, 400, 400),    # decoder_0 shortcut 1x1
]


@pytest.mark.parametrize("case", FULL_SIZE_CONVS)
def test_conv_adjoint_identities_full_size(dev, case):
    """Size-independent parity at the benchmark's real tensor sizes, where a CPU oracle would take minutes: the three conv
    kernels of a layer are tied together by the adjoint identities  <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)>
    (all three are the same trilinear form), evaluated with the same bf16 operands and fp32 accumulation; plus linearity of the
    forward in x.  Any tap/halo/edge/grouping error in one kernel breaks an identity."""
    from octave_amd import functional as F_
    Cin, Cout, k, s, p, g, B, H, W = case
    gen = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev).to(torch.bfloat16)
    x2 = torch.randn(B, Cin, H, W, generator=gen).to(dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * (1.0 / (k * k * Cin / g) ** 0.5)).to(dev)
    wq = torch.nn.Parameter(w.bfloat16().float().contiguous(memory_format=torch.channels_last))
    y = F_.raw_conv_fwd(x, wq, None, s, p, g)
    dy = torch.randn(tuple(y.shape), generator=gen).to(dev).to(torch.bfloat16)
    dyn = F_.to_nhwc(dy)
    dx = F_.raw_conv_dgrad(dyn, wq, tuple(x.shape), s, p, g)
    dw = F_.raw_conv_wgrad(x, dyn, wq, s, p, g)
    yf, dyf = F_.to_nchw_f32(y).double(), dy.double()
    t_fwd = (yf * dyf).sum().item()
    t_dgrad = (F_.to_nchw_f32(dx).double() * x.double()).sum().item()
    t_wgrad = (dw.double() * wq.detach().double()).sum().item()
    scale = (yf.abs() * dyf.abs()).sum().item()
    print(f"[adjoint {case}] fwd {t_fwd:.6e} dgrad {t_dgrad:.6e} wgrad {t_wgrad:.6e} (|terms| sum {scale:.3e})")
    # y and dx are rounded to bf16 on store (relative 2^-9 per element, random sign): the sums agree far better than that
    assert abs(t_fwd - t_dgrad) <= 2e-4 * scale and abs(t_fwd - t_wgrad) <= 2e-4 * scale, (t_fwd, t_dgrad, t_wgrad, scale)
    # linearity in x (exact in fp32 accumulation up to the bf16 rounding of the three outputs)
    y2 = F_.raw_conv_fwd(x2, wq, None, s, p, g)
    xs = (x.float() + x2.float()).to(torch.bfloat16)       # the sum is rounded: compare against conv of the ROUNDED sum's parts
    ys = F_.raw_conv_fwd(xs, wq, None, s, p, g)
    exact = (xs.float() - x.float() - x2.float()).abs().max().item()
    lin = (F_.to_nchw_f32(ys) - F_.to_nchw_f32(y) - F_.to_nchw_f32(y2)).abs().max().item()
    ymax = F_.to_nchw_f32(ys).abs().max().item()
    assert lin <= 0.03 * ymax + 8 * exact, (lin, ymax, exact)


def test_batch_norm_full_size(dev):
    """Train-mode BatchNorm at 16 x 64 x 400 x 400 (bf16), forward and backward against a float64 evaluation of the same
    formulas on the same bf16 inputs (torch ops on the GPU: the CPU oracle would need minutes at this size).  Outputs are
    stored in bf16, so the bound is one bf16 ulp of the value (2^-8 relative); with gamma = 1, beta = 0 the output channels
    must come out with mean 0 / variance 1.  (sum_i dx_i is NOT ~0 after the bf16 store: the float64 reference rounded to
    bf16 shows the same +-200 per channel.)"""
    from octave_amd import functional as F_
    B, C, H, W = 16, 64, 400, 400
    gen = torch.Generator(device="cpu").manual_seed(9)
    x = (torch.randn(B, C, H, W, generator=gen) * 3.0 + 1.5).to(dev).to(torch.bfloat16).requires_grad_(True)
    gam = torch.ones(C, device=dev, requires_grad=True)
    bet = torch.zeros(C, device=dev, requires_grad=True)
    y = F_.batch_norm(x, gam, bet, None, None, 0.1, 1e-5, True)
    yf = F_.to_nchw_f32(y.detach())
    m, v = yf.mean(dim=(0, 2, 3)), yf.var(dim=(0, 2, 3), unbiased=False)
    assert m.abs().max().item() < 2e-3 and (v - 1).abs().max().item() < 5e-3, (m.abs().max().item(), (v - 1).abs().max().item())
    g = torch.randn(B, C, H, W, generator=gen).to(dev).to(torch.bfloat16)
    (y.float() * g.float()).sum().backward()
    dx = F_.to_nchw_f32(x.grad).double()
    xd, gd = x.detach().double(), g.double()
    mu = xd.mean(dim=(0, 2, 3), keepdim=True)
    isd = 1.0 / torch.sqrt(xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
    xh = (xd - mu) * isd
    assert (yf.double() - xh).abs().max().item() <= 2.0 ** -8 * xh.abs().max().item() + 1e-6
    ref = isd * (gd - gd.mean(dim=(0, 2, 3), keepdim=True) - xh * (gd * xh).mean(dim=(0, 2, 3), keepdim=True))
    err = (dx - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all()), err.max().item()
    n = B * H * W
    assert (gam.grad.double() - (gd * xh).sum(dim=(0, 2, 3))).abs().max().item() < 1e-5 * n ** 0.5 + 1e-2
    assert (bet.grad.double() - gd.sum(dim=(0, 2, 3))).abs().max().item() < 1e-5 * n ** 0.5 + 1e-2


@pytest.mark.parametrize("shape", [(16, 1024, 25, 25), (16, 256, 50, 50), (16, 2048, 13, 13), (5, 72, 31, 17)])
def test_batch_norm_fused_finalize_repeated(dev, shape):
    """The small-tensor path (two launches: the apply kernels merge the <= 32 slab partials of their own channels; DESIGN.md
    3.5) on shapes with many row slabs per column block, plus one whose 9 channel chunks send it down the three-launch path (ragged
    last slab), called three times in a row, forward and backward against a float64 evaluation of the same formulas on the
    same bf16 inputs; running statistics included."""
    from octave_amd import functional as F_
    B, C, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(77)
    x0 = (torch.randn(B, C, H, W, generator=gen) * 2.0 + 0.7).to(dev).to(torch.bfloat16)
    g = torch.randn(B, C, H, W, generator=gen).to(dev).to(torch.bfloat16)
    xd, gd = x0.double(), g.double()
    mu = xd.mean(dim=(0, 2, 3), keepdim=True)
    var = xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    isd = 1.0 / torch.sqrt(var + 1e-5)
    xh = (xd - mu) * isd
    ref_dx = isd * (gd - gd.mean(dim=(0, 2, 3), keepdim=True) - xh * (gd * xh).mean(dim=(0, 2, 3), keepdim=True))
    n = B * H * W
    for it in range(3):
        x = x0.clone().requires_grad_(True)
        gam = torch.ones(C, device=dev, requires_grad=True)
        bet = torch.zeros(C, device=dev, requires_grad=True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y = F_.batch_norm(x, gam, bet, rm, rv, 0.1, 1e-5, True)
        yf = F_.to_nchw_f32(y.detach()).double()
        assert (yf - xh).abs().max().item() <= 2.0 ** -8 * xh.abs().max().item() + 1e-6, it
        assert (rm.double() - 0.1 * mu.flatten()).abs().max().item() < 1e-5, it
        assert (rv.double() - (0.9 + 0.1 * var.flatten() * n / (n - 1))).abs().max().item() < 1e-4, it
        (y.float() * g.float()).sum().backward()
        err = (F_.to_nchw_f32(x.grad).double() - ref_dx).abs()
        assert bool((err <= 2.0 ** -7 * ref_dx.abs() + 2e-5).all()), (it, err.max().item())
        assert (gam.grad.double() - (gd * xh).sum(dim=(0, 2, 3))).abs().max().item() < 1e-5 * n ** 0.5 + 1e-2, it
        assert (bet.grad.double() - gd.sum(dim=(0, 2, 3))).abs().max().item() < 1e-5 * n ** 0.5 + 1e-2, it


# ----------------------------------------------------------------------------------------- round 2: 8-wave kernels
IGEMM8_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, groups
    (2, 64, 17, 19, 136, 3, 1, 1, 1),      # odd image, N tail, 3x3
    (3, 128, 12, 10, 320, 1, 1, 0, 1),     # 1x1, two K steps, N = 256 + tail
    (2, 64, 15, 16, 128, 3, 2, 1, 1),      # stride 2 (forward only on the 8-wave kernel; dgrad falls back)
    (2, 128, 9, 9, 256, 3, 1, 1, 2),       # grouped (Cg = 64)
    (1, 192, 20, 20, 130, 2, 2, 0, 1),     # k2 s2 (the ConvTranspose adjoint shape)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("algo", [2, 3, 8])
@pytest.mark.parametrize("case", IGEMM8_CASES)
def test_igemm8_forward_and_dgrad_vs_torch(dev, case, algo, dtype):
    """The 8-wave LDS-DMA kernel (octa_conv_desc.algo 2 / 3) on awkward shapes against torch's CPU conv on the same rounded
    operands (fp32 accumulation on both sides)."""
    from octave_amd import functional as F_
    B, Cin, H, W, Cout, k, s, p, g = case
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias, s, p, 1, g))
    F_._ALGO_OVERRIDE = algo
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), s, p, g, 1)
        from octave_amd._lib import lib
        name = lib().octa_last_conv_kernel().decode()
        assert "conv_igemm8_kernel" in name, name
        t = TOL[dtype]
        check(f"igemm8 fwd {case} algo {algo}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))
        dy = torch.randn(tuple(want.shape), generator=gen).to(dtype).float()
        dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), s, p, g)
    finally:
        F_._ALGO_OVERRIDE = 0
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w, None, s, p, 1, g).backward(dy)
    check(f"igemm8 dgrad {case} algo {algo}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))


PWGEMM_CASES = [
    # B, Cin, H, W, Cout : 1x1 / stride 1 / no padding (the persistent pointwise GEMM, pwgemm.hpp)
    (3, 128, 12, 10, 320),     # two K stages, N = 256 + tail, M = 360 (tile tails on both axes)
    (2, 64, 37, 23, 72),       # one K stage per tile, odd image, N tail inside one tile
    (5, 256, 17, 19, 576),     # 13 M tiles x 3-5 N tiles; the data gradient runs 9 K stages
    (1, 192, 9, 7, 192),       # a single M tile, three K stages, N = 1.5 tiles
    (2, 512, 33, 31, 264),     # eight K stages, N = 256 + 8 (the data gradient, K = 264, falls back)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("algo", [9, 10, 11])
@pytest.mark.parametrize("case", PWGEMM_CASES)
def test_pwgemm_forward_dgrad_addend_vs_torch(dev, case, algo, dtype):
    """pwgemm.hpp (octa_conv_desc.algo 9 / 10 / 11: 256 x 128, 128 x 256, 128 x 128 tiles): forward (+ bias, ReLU), data gradient
    and data gradient with the fused addend against torch's CPU conv on the same rounded operands; tile tails on M and N, one to
    eight K stages per tile, several tiles per workgroup."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout = case
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin, 1, 1, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias))
    t = TOL[dtype]
    F_._ALGO_OVERRIDE = algo
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, 0, 1, 1)
        name = lib().octa_last_conv_kernel().decode()
        assert "pwgemm_kernel" in name, name
        check(f"pwgemm fwd {case} algo {algo}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))
        dy = torch.randn(tuple(want.shape), generator=gen).to(dtype).float()
        dyd = F_.to_nhwc(dy.to(dev), dtype=dtype)
        dx = F_.raw_conv_dgrad(dyd, wd, (B, Cin, H, W), 1, 0, 1)
        assert ("pwgemm_kernel" in lib().octa_last_conv_kernel().decode()) == (Cout % 64 == 0)        # K = Cout must be whole 64-channel stages
        add = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
        dxa = F_.raw_conv_dgrad(dyd, wd, (B, Cin, H, W), 1, 0, 1, addend=F_.to_nhwc(add.to(dev), dtype=dtype))
        assert ("pwgemm_kernel" in lib().octa_last_conv_kernel().decode()) == (Cout % 64 == 0)
    finally:
        F_._ALGO_OVERRIDE = 0
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w, None).backward(dy)
    check(f"pwgemm dgrad {case} algo {algo}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))
    check(f"pwgemm dgrad+addend {case} algo {algo}", dxa, xr.grad + add, t["rtol"], t["atol"] * float((xr.grad + add).abs().max()))


@pytest.mark.parametrize("algo", [9, 10, 11])
def test_pwgemm_conv_transpose_upshuffle_vs_torch(dev, algo):
    """The conv-transpose k2 s2 up-shuffle (a pointwise GEMM with N = 4 Cout scattered to (2i + di, 2j + dj)) through pwgemm."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout = 2, 128, 11, 13, 72
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(B, Cin, H, W, generator=gen).bfloat16().float()
    w = (torch.randn(Cin, Cout, 2, 2, generator=gen) * 0.1).bfloat16().float()
    bias = torch.randn(Cout, generator=gen)
    want = torch.nn.functional.conv_transpose2d(x, w, bias, stride=2)
    F_._ALGO_OVERRIDE = algo
    try:
        with torch.no_grad():
            y = F_.conv_transpose2x2(x.to(dev).to(torch.bfloat16), w.to(dev), bias.to(dev))
        assert "pwgemm_kernel" in lib().octa_last_conv_kernel().decode(), lib().octa_last_conv_kernel().decode()
    finally:
        F_._ALGO_OVERRIDE = 0
    t = TOL[torch.bfloat16]
    check(f"pwgemm up-shuffle algo {algo}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 128, 11, 13, 72, 64), (1, 64, 25, 25, 256, 256), (2, 256, 6, 9, 64, 64)])
def test_upsample_cat_matches_cat_of_conv_transpose(dev, case, dtype):
    """functional.upsample_cat (round 5: the up-shuffle GEMM stores into its channel slice of the cat buffer) against
    torch.cat((skip, conv_transpose2d(x))) on CPU, and against the two separate ops of this library (cat_crop of conv_transpose2x2:
    same kernels, only the output's per-pixel stride differs -> bit-identical forward and data gradients)."""
    from octave_amd import functional as F_
    B, Cin, H, W, Cout, Cs = case
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    skip = torch.randn(B, Cs, 2 * H, 2 * W, generator=gen).to(dtype).float()
    w = (torch.randn(Cin, Cout, 2, 2, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    dout = torch.randn(B, Cs + Cout, 2 * H, 2 * W, generator=gen).to(dtype).float()
    xr, sr, wr, br = (t.clone().requires_grad_(True) for t in (x, skip, w, bias))
    want = torch.cat((sr, F.conv_transpose2d(xr, wr, br, stride=2)), dim=1)
    want.backward(dout)

    def run(fused):
        xs = F_.to_nhwc(x.to(dev), dtype=dtype).requires_grad_(True)
        ss = F_.to_nhwc(skip.to(dev), dtype=dtype).requires_grad_(True)
        wd, bd = w.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
        old = F_._FUSE_UPCAT
        F_._FUSE_UPCAT = fused
        try:
            y = F_.upsample_cat(ss, xs, wd, bd)
        finally:
            F_._FUSE_UPCAT = old
        assert (type(y.grad_fn).__name__ == "UpCatFnBackward") == fused, type(y.grad_fn).__name__
        y.backward(F_.to_nhwc(dout.to(dev), dtype=dtype))
        return y, xs.grad, ss.grad, wd.grad, bd.grad
    got, twin = run(True), run(False)
    t = TOL[dtype]
    for name, g, tw, wnt in zip(("y", "dx", "dskip", "dw", "dbias"), got, twin, (want, xr.grad, sr.grad, wr.grad, br.grad)):
        check(f"upsample_cat {name} {case}", g, wnt, t["rtol"], t["atol"] * float(wnt.abs().max()))
        if name in ("y", "dx", "dskip"):        # (the weight / bias gradients may add their partial sums in launch order)
            assert torch.equal(g.float().cpu(), tw.float().cpu()), f"upsample_cat {name}: fused and separate ops differ"


HALO8_CASES = [
    # B, Cin, H, W, Cout, groups : 3x3 / stride 1 / pad 1 (halo8.hpp)
    (2, 64, 17, 19, 136, 1),       # odd image, N tail, one 64-channel slice
    (1, 128, 26, 50, 256, 1),      # two slices, several patches per image (partial patches at the right / bottom edge)
    (2, 128, 9, 9, 256, 2),        # grouped: Cg = 64, Ng = 128
    (3, 192, 25, 25, 320, 1),      # three slices, 25 x 25 (the decoder_4 / encoder_3 image), N = 256 + 64
    (1, 256, 13, 13, 512, 4),      # four groups of 64 -> 128
    (1, 64, 40, 7, 72, 1),         # narrow image: patch narrower than any default
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("algo", [12, 13, 14, 112])
@pytest.mark.parametrize("case", HALO8_CASES)
def test_halo8_forward_and_dgrad_vs_torch(dev, case, algo, dtype):
    """halo8.hpp (octa_conv_desc.algo 12): the 8-wave 3x3 kernel with a 2-D pixel patch per tile, forward (+ bias, ReLU) and
    data gradient (flipped taps) against torch's CPU conv on the same rounded operands.  13 = its v_mfma_f32_16x16x32 form
    (halo16.hpp); 112 = algo 12 with the linear patch image of round 4 (octa_tuning_set(6, 0)) instead of the packed,
    bank-conflict-free one; 14 = halo16 with the persistent tile loop (halo16p.hpp: several tiles per workgroup on the small cases'
    grids only when the tile count exceeds the CU count -- test_halo16p_many_tiles_per_workgroup covers that)."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, g = case
    gen = torch.Generator().manual_seed(19)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, 3, 3, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias, 1, 1, 1, g))
    t = TOL[dtype]
    kname = {13: "conv_halo16_kernel", 14: "conv_halo16p_kernel"}.get(algo, "conv_halo8_kernel")
    lib().octa_tuning_set(6, 0 if algo == 112 else 1)
    F_._ALGO_OVERRIDE = 12 if algo == 112 else algo
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, 1, g, 1)
        name = lib().octa_last_conv_kernel().decode()
        assert kname in name, name
        check(f"halo8 fwd {case} algo {algo}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))
        dy = torch.randn(tuple(want.shape), generator=gen).to(dtype).float()
        dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), 1, 1, g)
        assert (kname in lib().octa_last_conv_kernel().decode()) == ((Cout // g) % 64 == 0)
    finally:
        F_._ALGO_OVERRIDE = 0
        lib().octa_tuning_set(6, 1)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w, None, 1, 1, 1, g).backward(dy)
    check(f"halo8 dgrad {case} algo {algo}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("algo", [12, 13, 14])
@pytest.mark.parametrize("case", [(2, 256, 20, 20, 136, 1), (1, 512, 25, 25, 256, 2), (3, 384, 9, 30, 128, 1)])
def test_halo8_tail_split_vs_torch(dev, case, dtype, algo):
    """halo8 with the tail-split scratch registered (octa_conv_desc.ws): a handful of tiles on 256 CUs, so every tile is
    split over the 64-channel slices into 2-3 parts of raw fp32 partial tiles that halo8_splitk_fix_kernel sums, biases, activates
    and stores at the patch's pixels; forward and data gradient against torch's CPU conv."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, g = case
    gen = torch.Generator().manual_seed(29)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, 3, 3, generator=gen) * 0.05).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias, 1, 1, 1, g))
    t = TOL[dtype]
    ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)
    F_.set_splitk_workspace(ws)
    F_._ALGO_OVERRIDE = algo
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, 1, g, 1)
        name = lib().octa_last_conv_kernel().decode()
        assert {13: "conv_halo16_kernel", 14: "conv_halo16p_kernel"}.get(algo, "conv_halo8_kernel") in name and "+tail" in name, name
        check(f"halo8 split fwd {case}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))
        dy = torch.randn(tuple(want.shape), generator=gen).to(dtype).float()
        dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), 1, 1, g)
        dname = lib().octa_last_conv_kernel().decode()
    finally:
        F_._ALGO_OVERRIDE = 0
        F_.set_splitk_workspace(None)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w, None, 1, 1, 1, g).backward(dy)
    check(f"halo8 split dgrad {case} [{dname}]", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))


@pytest.mark.parametrize("case", [(16, 128, 50, 50, 256, 1, False), (9, 256, 50, 50, 384, 1, True), (8, 256, 60, 44, 512, 4, False)])
def test_halo16p_many_tiles_per_workgroup(dev, case):
    """halo16p.hpp (algo 14): grids of several hundred tiles, so every workgroup walks 2-4 work items back to back (whole tiles, then
    with the scratch the parts of the tail-split tiles; grouped layer included): the item hand-over -- next item's patch / weight prefetch
    during the last slice, fragment reads across the item change, epilogue between two items -- against torch's CPU conv, forward and
    data gradient, and bit-identical to the one-tile-per-workgroup kernel (algo 13: same MFMA order per tile)."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, g, split = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, 3, 3, generator=gen) * 0.05).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias, 1, 1, 1, g))
    t = TOL[dtype]
    ws = torch.empty(32 << 20, dtype=torch.float32, device=dev) if split else None
    F_.set_splitk_workspace(ws)
    outs = {}
    try:
        for algo in (14, 13):
            F_._ALGO_OVERRIDE = algo
            y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, 1, g, 1)
            name = lib().octa_last_conv_kernel().decode()
            assert ("conv_halo16p_kernel" if algo == 14 else "conv_halo16_kernel") in name and (("+tail" in name) == split), name
            dy = torch.randn(tuple(want.shape), generator=torch.Generator().manual_seed(43)).to(dtype).float()
            dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), 1, 1, g)
            outs[algo] = (y, dx)
    finally:
        F_._ALGO_OVERRIDE = 0
        F_.set_splitk_workspace(None)
    check(f"halo16p fwd {case}", outs[14][0], want, t["rtol"], t["atol"] * float(want.abs().max()))
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w, None, 1, 1, 1, g).backward(dy)
    check(f"halo16p dgrad {case}", outs[14][1], xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))
    assert torch.equal(outs[14][0], outs[13][0]) and torch.equal(outs[14][1], outs[13][1]), "persistent and one-tile-per-workgroup kernels differ"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_spectral_norm_batch_writes_the_packed_conv_operands(dev, dtype):
    """octa_sn_job.packed_fwd / packed_dgrad_taps (round 5): the launch that writes w / sigma also writes the conv's packed operands.
    They must equal, bit for bit, what the stand-alone pack kernels make of the returned normalised weight, and a conv through
    functional.conv2d must pick them up (no pack launch: the weight is a plain tensor, so nothing else would cache it)."""
    from octave_amd import functional as F_
    gen = torch.Generator().manual_seed(5)
    triples = []
    for (Cout, Cin) in [(128, 15), (256, 15)]:
        w = torch.nn.Parameter((torch.randn(Cout, Cin, 4, 4, generator=gen) * 0.1).to(dev))
        u = torch.nn.functional.normalize(torch.randn(Cout, generator=gen), dim=0).to(dev)
        v = torch.nn.functional.normalize(torch.randn(Cin * 16, generator=gen), dim=0).to(dev)
        triples.append((w, u, v))
    wn = F_.spectral_norm_batch(triples, True, 1e-12, dtype)
    for (w, u, v), wsn in zip(triples, wn):
        Cout, Cin = w.shape[:2]
        pre = getattr(wsn, "_octa_packed", None)
        assert pre is not None and ("fwd", dtype, 1, 16) in pre and ("dgrad_taps", dtype, 1, F_.round8(Cout)) in pre
        plain = wsn.detach().clone()                         # same values, no operands attached
        for kind, pad in (("fwd", 16), ("dgrad_taps", F_.round8(Cout))):
            want = F_._packed(plain, kind, dtype, 1, pad)
            assert torch.equal(pre[(kind, dtype, 1, pad)].view(torch.uint8), want.view(torch.uint8)), (kind, Cout)
        assert F_._packed(wsn, "fwd", dtype, 1, 16) is pre[("fwd", dtype, 1, 16)]


def test_probe_stream_load_leaves_the_buffer_unchanged(dev):
    """octa_probe_stream_load (tools/comm_pressure.py's stand-in for a collective's kernels): any footprint / repetition count, the
    buffer's contents are what they were."""
    from octave_amd._lib import lib
    x = torch.randn(1 << 20, device=dev)
    want = x.clone()
    for nb, reps in ((1, 1), (32, 3), (256, 2)):
        lib().octa_probe_stream_load(x.data_ptr(), x.numel() * 4, nb, reps, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(x, want)


def test_two_streams_split_convs_with_their_own_scratch(dev):
    """SURVEY 8(b) "re-entrant across streams": the tail-split scratch travels in the call (octa_conv_desc.ws), so two streams run
    split convs CONCURRENTLY, each with a buffer of its own, with no registration dance: 20 alternating launches per stream of two
    different layers, every result equal to the same layer run alone (bit for bit: the split is deterministic)."""
    import ctypes
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    L = lib()
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(31)
    layers = []
    for (B, Cin, H, W, Cout) in [(2, 256, 20, 20, 136), (1, 512, 25, 25, 256)]:
        x = F_.to_nhwc((torch.randn(B, Cin, H, W, generator=gen)).to(dev), dtype=dtype)
        w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05).to(dev).contiguous(memory_format=torch.channels_last))
        wp = F_._packed(w, "fwd", dtype, 1, F_.round8(Cin))
        y = F_.nhwc_empty(B, Cout, H, W, dtype, dev)
        d = F_._desc(B, H, W, H, W, Cin, Cout, 3, 3, 1, 1, 1, F_.nhwc_ld(x), F_.nhwc_ld(y), dtype)
        d.algo = 12
        ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
        layers.append((d, x, wp, y, ws, w))
    torch.cuda.synchronize()
    alone = []
    for d, x, wp, y, ws, _ in layers:
        L.octa_conv2d_fwd(ctypes.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert "+tail" in L.octa_last_conv_kernel().decode(), L.octa_last_conv_kernel().decode()
        torch.cuda.synchronize()
        alone.append(y.clone())
        y.zero_()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [[], []]
    for it in range(20):
        for si, (d, x, wp, y, ws, _) in enumerate(layers):
            with torch.cuda.stream(streams[si]):
                L.octa_conv2d_fwd(ctypes.byref(d), x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), streams[si].cuda_stream)
                outs[si].append(y.clone())
    torch.cuda.synchronize()
    for si in range(2):
        for it, o in enumerate(outs[si]):
            assert torch.equal(o, alone[si]), f"stream {si} launch {it}: differs from the layer run alone"


ADD_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, groups
    (2, 256, 13, 11, 64, 1, 1, 0, 1),      # a bottleneck's conv1 (1x1): dx has 256 channels
    (2, 136, 9, 10, 64, 1, 1, 0, 1),       # N tail
    (2, 64, 12, 12, 64, 3, 1, 1, 1),       # 3x3 (the halo / resident kernels must step aside)
    (1, 20, 7, 9, 24, 1, 1, 0, 1),         # odd channel counts: scalar addend reads
    (2, 64, 16, 16, 128, 3, 2, 1, 1),      # strided: GEMM + col2im path, the sum is a separate add
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("algo", [0, 1, 5, 2, 8, 7])
@pytest.mark.parametrize("case", ADD_CASES)
def test_dgrad_with_fused_addend_vs_torch(dev, case, algo, dtype):
    """octa_conv2d_dgrad_add: dx = conv^T(dy, w) + addend in the kernel's epilogue, on every kernel family the dispatcher may
    pick (algo 7 must fall back: the resident-weight kernel has no addend path), against torch's CPU gradient + addend."""
    from octave_amd import functional as F_
    B, Cin, H, W, Cout, k, s_, p_, g = case
    gen = torch.Generator().manual_seed(31)
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * 0.1).to(dtype).float()
    xr = torch.randn(B, Cin, H, W, generator=gen).requires_grad_(True)
    y = torch.nn.functional.conv2d(xr, w, None, s_, p_, 1, g)
    dy = torch.randn(tuple(y.shape), generator=gen).to(dtype).float()
    y.backward(dy)
    add = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    want = xr.grad + add
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    F_._ALGO_OVERRIDE = algo
    try:
        dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), s_, p_, g,
                               addend=F_.to_nhwc(add.to(dev), dtype=dtype))
    finally:
        F_._ALGO_OVERRIDE = 0
    t = TOL[dtype]
    check(f"dgrad+addend {case} algo {algo}", dx, want, t["rtol"], t["atol"] * float(want.abs().max()))


def test_fanout_gradient_fusion_matches_autograd_sum(dev):
    """functional.stash_grad + Conv2dFn's grad_holder (Bottleneck.forward): the gradient of a tensor feeding a conv AND a
    shortcut equals autograd's own sum, including when the tensor has a third consumer and when the holder goes unused."""
    from octave_amd import functional as F_
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 64, 10, 10, generator=gen)
    w = torch.nn.Parameter((torch.randn(32, 64, 1, 1, generator=gen) * 0.1).to(dev).contiguous(memory_format=torch.channels_last))
    gy = torch.randn(2, 32, 10, 10, generator=gen).to(dev)

    def run(fused):
        x = F_.to_nhwc(x0.to(dev), dtype=torch.bfloat16).requires_grad_(True)
        t = x * 1.0                                     # non-leaf, like a block input
        h = F_.GradHolder() if fused else None
        y = F_.conv2d(t, w, None, 1, 0, 1, 0, h)
        ts = F_.stash_grad(t, h) if fused else t
        loss = (y.float() * gy).sum() + (ts.float() * 0.5).sum() + (t.float() * 0.25).sum()    # conv + shortcut + third consumer
        loss.backward()
        return x.grad.float().cpu()

    a, b = run(False), run(True)
    assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()), float((a - b).abs().max())
    # a holder nobody stashes into: plain data gradient
    x = F_.to_nhwc(x0.to(dev), dtype=torch.bfloat16).requires_grad_(True)
    (F_.conv2d(x, w, None, 1, 0, 1, 0, F_.GradHolder()).float() * gy).sum().backward()
    x2 = F_.to_nhwc(x0.to(dev), dtype=torch.bfloat16).requires_grad_(True)
    (F_.conv2d(x2, w, None, 1, 0, 1, 0).float() * gy).sum().backward()
    assert torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 15, 40, 44, 128, 4, 2, 1), (3, 2, 50, 38, 64, 4, 2, 1), (2, 13, 21, 21, 72, 4, 2, 1)])
def test_strided_dgrad_tap_major_col2im_vs_torch(dev, case, dtype):
    """Stride-2 data gradient of the few-channel discriminator convs: GEMM with the tap-major operand + octa_col2im_taps
    (16-byte channel vectors per tap) against torch's CPU gradient; the padding channels of dx come back as zeros."""
    from octave_amd import functional as F_
    B, Cin, H, W, Cout, k, s_, p_ = case
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = (torch.randn(Cout, Cin, k, k, generator=gen) * 0.1).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xr, w, None, s_, p_)
    dy = torch.randn(tuple(y.shape), generator=gen).to(dtype).float()
    y.backward(dy)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)      # a plain tensor, like the spectral-normalised weights
    dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), s_, p_, 1)
    t = TOL[dtype]
    check(f"tap-major dgrad {case}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))
    ld = F_.nhwc_ld(dx)
    full = torch.as_strided(dx.permute(0, 2, 3, 1), (B, H, W, ld), (H * W * ld, W * ld, ld, 1))
    assert float(full[..., Cin:].float().abs().max()) == 0.0, "padding channels not zero"


RES_CASES = [
    # B, Cin, H, W, Cout, k, pad : the resident-weight persistent kernel (algo 7); enough tiles for several per workgroup,
    # ragged right / bottom tiles, every channel-count bucket (13 -> zero-filled padding channels)
    (3, 32, 150, 170, 64, 3, 1),
    (2, 64, 100, 97, 32, 3, 1),
    (2, 32, 67, 300, 24, 3, 1),
    (2, 64, 150, 131, 13, 1, 0),
    (2, 64, 130, 120, 256, 1, 0),      # 16 n-tiles: two passes of 8 over the resident patch
    (2, 64, 110, 90, 128, 1, 0),       # 8 n-tiles, one pass
    (2, 64, 75, 101, 200, 1, 0),       # two passes, ragged channel tail
    (3, 32, 140, 140, 64, 1, 0),
    (2, 64, 90, 200, 40, 1, 0),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("grid", [0, 7])
@pytest.mark.parametrize("case", RES_CASES)
def test_resident_weight_conv_vs_torch(dev, case, grid, dtype, monkeypatch):
    """convres.hpp (octa_conv_desc.algo 7): forward (+bias, ReLU) and data gradient against torch's CPU conv on the same
    rounded operands; the padding channels of a 13-channel output must come back as zeros (zero_pad)."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, k, p = case
    if grid:
        monkeypatch.setenv("OCTA_CONVRES_GRID", str(grid))      # 7 workgroups: dozens of tiles each through the patch ring
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin, k, k, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    want = torch.relu(torch.nn.functional.conv2d(x, w, bias, 1, p))
    t = TOL[dtype]
    F_._ALGO_OVERRIDE = 7
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, p, 1, 1)
        name = lib().octa_last_conv_kernel().decode()
        assert "conv_res" in name, name
        check(f"res fwd {case}", y, want, t["rtol"], t["atol"] * float(want.abs().max()))
        ld = F_.nhwc_ld(y)
        if ld != Cout:
            buf = y.permute(0, 2, 3, 1)
            full = torch.as_strided(buf, (B, H, W, ld), (H * W * ld, W * ld, ld, 1))
            assert float(full[..., Cout:].float().abs().max()) == 0.0, "padding channels not zero-filled"
        # the data gradient of a conv with Cout in {32, 64} gathers dy with 32 / 64 channels: the same kernel family
        dx = None
        if Cout in (32, 64):
            dy = torch.randn(tuple(want.shape), generator=gen).to(dtype).float()
            dx = F_.raw_conv_dgrad(F_.to_nhwc(dy.to(dev), dtype=dtype), wd, (B, Cin, H, W), 1, p, 1)
            assert "conv_res" in lib().octa_last_conv_kernel().decode()
    finally:
        F_._ALGO_OVERRIDE = 0
    if dx is not None:
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.conv2d(xr, w, None, 1, p).backward(dy)
        check(f"res dgrad {case}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))


FOLD_CASES = [
    # Cin, Cout, k, s, p, g, B, H, W, bias : the few-channel weight gradients (partial tiles + fold, the fold scratch)
    (2, 64, 4, 2, 1, 1, 2, 96, 96, True),       # discriminator stack_0
    (3, 32, 3, 2, 1, 1, 2, 80, 72, False),      # stem
    (64, 13, 1, 1, 0, 1, 2, 50, 44, True),      # squeeze conv (13 output channels)
    (15, 128, 4, 2, 1, 1, 2, 48, 48, True),     # spectral conv, padded input channels
    (64, 128, 3, 1, 1, 2, 2, 40, 36, False),    # grouped (encoder_2's split-attention conv)
    (64, 32, 1, 1, 0, 1, 1, 120, 100, False),   # decoder_0 shortcut
    (64, 32, 3, 1, 1, 1, 1, 130, 140, True),    # halo weight-gradient kernel <2> (bf16): one slice per tile range
    (32, 64, 3, 1, 1, 4, 1, 130, 140, False),   # halo kernel, one dense set of four groups (block diagonal)
    (64, 128, 3, 1, 1, 4, 1, 130, 140, True),   # halo kernel, two channel sets of paired groups
    (128, 64, 3, 1, 1, 1, 1, 128, 136, False),  # halo kernel <4>, four channel-chunk blocks per tile range
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", FOLD_CASES)
def test_wgrad_partial_tiles_and_fold_vs_torch(dev, case, dtype):
    """conv_wgrad_kernel in partial-store mode + wgrad_fold_kernel against torch's CPU gradient on the same rounded operands:
    += semantics (the gradient buffer starts non-zero), bias gradient, padded channels, groups; bit-identical from run to run
    (fixed summation order), and within rounding of the float-atomic epilogue it replaces."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    Cin, Cout, k, s_, p_, g, B, H, W, with_bias = case
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * 0.1)
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    br = torch.zeros(Cout, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, wr, br, s_, p_, 1, g)
    dy = torch.randn(tuple(y.shape), generator=gen).to(dtype).float()
    y.backward(dy)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    dyd = F_.to_nhwc(dy.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)

    def run(fold):
        F_.set_wgrad_fold_workspace(ws if fold else None)
        try:
            dw = torch.full((Cout, k, k, Cin // g), 0.25, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
            db = torch.full((Cout,), -0.5, dtype=torch.float32, device=dev) if with_bias else None
            F_.raw_conv_wgrad(xd, dyd, wd, s_, p_, g, dw=dw, dbias=db)
            name = lib().octa_last_conv_kernel().decode()
        finally:
            F_.set_wgrad_fold_workspace(None)
        return dw, db, name
    dw1, db1, n1 = run(True)
    dw2, db2, n2 = run(True)
    dw0, db0, n0 = run(False)
    assert n1.endswith("+fold") and "wgrad" in n1 and not n0.endswith("+fold"), (n1, n0)
    if k == 3 and H * W >= 128 * 128 and dtype == torch.bfloat16:
        assert "wgrad_halo" in n1, n1
    assert torch.equal(dw1, dw2) and (db1 is None or torch.equal(db1, db2)), "fold is not deterministic"
    t = TOL[dtype]
    sc = float(wr.grad.abs().max())
    check(f"fold wgrad {case}", dw1 - 0.25, wr.grad, t["rtol"], t["atol"] * sc * 4)
    check(f"fold vs atomics {case}", dw1, dw0, 1e-4, 1e-4 * sc)
    if with_bias:
        check(f"fold bias grad {case}", db1 + 0.5, br.grad, t["rtol"], t["atol"] * float(br.grad.abs().max()) * 4)


GROUPED_RES_CASES = [
    # B, Cin, H, W, Cout, k, pad, groups : grouped layers on the resident-weight kernel (blockIdx.y = group; round 4)
    (2, 64, 100, 97, 128, 3, 1, 2),     # encoder_2's split-attention conv: 32 -> 64 per group; its data gradient gathers 64 -> 32
    (2, 128, 70, 90, 128, 3, 1, 4),     # 32 -> 32 per group, four groups
    (2, 128, 90, 110, 64, 1, 0, 2),     # pointwise, 64 -> 32 per group
    (1, 64, 130, 140, 128, 3, 1, 4),    # decoder_1's split-attention conv (16 -> 32 per group): pairs of groups merged into 32 -> 64 blocks
    (2, 64, 100, 97, 64, 3, 1, 1),      # 64 -> 64: the compact-stage instantiation (72 KB of resident weights + two 41 KB patches)
    (2, 256, 60, 70, 512, 3, 1, 4),     # decoder_2's split-attention conv: 64 -> 128 per group = two 64-channel output slices per group
    (1, 128, 70, 90, 256, 3, 1, 2),     # encoder_3's: 64 -> 128 per group, two groups
    (1, 64, 50, 70, 256, 3, 1, 1),      # 64 -> 256 ungrouped: four output slices
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("grid", [0, 3])
@pytest.mark.parametrize("case", GROUPED_RES_CASES)
def test_grouped_resident_weight_conv_vs_torch(dev, case, grid, dtype, monkeypatch):
    """Grouped forward (+bias) / data gradient on convres.hpp (algo 7) against torch's CPU conv on the same rounded operands,
    and -- for the 16 -> 32-per-group layer -- the weight gradient of the pair-merged layer (conv3x3_wgrad_halo with channel sets)."""
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, k, p, g = case
    if grid:
        monkeypatch.setenv("OCTA_CONVRES_GRID", str(grid))
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * 0.1).to(dtype).float()
    bias = torch.randn(Cout, generator=gen)
    dy = torch.randn(B, Cout, H, W, generator=gen).to(dtype).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    want = torch.nn.functional.conv2d(xr, wr, bias, 1, p, 1, g)
    want.backward(dy)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    wd = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last))
    dyd = F_.to_nhwc(dy.to(dev), dtype=dtype)
    t = TOL[dtype]
    merged = Cin // g == 16
    assert bool(F_._densify(g, Cin, Cout, k, k, 1, p, H, W, dtype)) == merged
    F_._ALGO_OVERRIDE = 7
    try:
        y = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, p, g)
        assert "conv_res" in lib().octa_last_conv_kernel().decode(), lib().octa_last_conv_kernel().decode()
        dx = F_.raw_conv_dgrad(dyd, wd, (B, Cin, H, W), 1, p, g)
        if Cout // g in (32, 64) or merged:     # (the data gradient gathers Cout / g channels: resident-weight shapes are 32 / 64)
            assert "conv_res" in lib().octa_last_conv_kernel().decode(), lib().octa_last_conv_kernel().decode()
    finally:
        F_._ALGO_OVERRIDE = 0
    check(f"grouped res fwd {case}", y, want.detach(), t["rtol"], t["atol"] * float(want.detach().abs().max()))
    check(f"grouped res dgrad {case}", dx, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))
    if merged:
        # the same layer through the library's own kernel choice (no override), and its weight gradient
        y0 = F_.raw_conv_fwd(xd, wd, bias.to(dev), 1, p, g)
        dx0 = F_.raw_conv_dgrad(dyd, wd, (B, Cin, H, W), 1, p, g)
        check(f"merged fwd {case}", y0, want.detach(), t["rtol"], t["atol"] * float(want.detach().abs().max()))
        check(f"merged dgrad {case}", dx0, xr.grad, t["rtol"], t["atol"] * float(xr.grad.abs().max()))
        dw = F_.raw_conv_wgrad(xd, dyd, wd, 1, p, g)
        if dtype == torch.bfloat16:
            assert "wgrad_halo" in lib().octa_last_conv_kernel().decode(), lib().octa_last_conv_kernel().decode()
        check(f"merged wgrad {case}", dw, wr.grad, t["rtol"], t["atol"] * float(wr.grad.abs().max()) * 4)


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_resident_weight_upshuffle_vs_torch(dev, dtype):
    """ConvTranspose2d k2 s2 (64 -> 64) as the 1x1 'upshuffle' GEMM on the resident-weight kernel."""
    import octave_amd.layers as L_
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    gen = torch.Generator().manual_seed(5)
    m = L_.ConvTranspose2d(64, 64, 2, 2).to(dev)
    x = torch.randn(2, 64, 120, 136, generator=gen).to(dtype)
    ref = torch.nn.functional.conv_transpose2d(x.float(), m.weight.detach().cpu().to(dtype).float(), m.bias.detach().cpu(), 2)
    F_._ALGO_OVERRIDE = 7
    try:
        with torch.no_grad():
            y = m(F_.to_nhwc(x.to(dev), dtype=dtype))
        assert "conv_res1x1" in lib().octa_last_conv_kernel().decode(), lib().octa_last_conv_kernel().decode()
    finally:
        F_._ALGO_OVERRIDE = 0
    t = TOL[dtype]
    check("res upshuffle", y, ref, t["rtol"], t["atol"] * float(ref.abs().max()))


@pytest.mark.parametrize("sched", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad_batch_vs_torch(dev, dtype, fold, sched):
    """octa_conv2d_wgrad_batch: a mixed queue (both slab orientations of wgrad8, the 256 x 256 tiles of wgrad9, grouped convs,
    strided ones, a small-N job that falls through to the single-problem kernel, fused bias gradients) in ONE call against
    torch's CPU gradients.  fold: with the fold scratch registered the M-split jobs of every kernel family store
    partial tiles and the batch ends in fold launches (two of them: more than 16 jobs would need a third); the gradients must
    also be bit-identical between two runs.  sched: the schedule of the 256 x 256 kernel (octa_tuning_set(8, .)): 0 = rounds of one
    split length (wgrad9), 1 = per-class splits on XCD-interleaved sequences (wgrad9x), 2 = the same, persistent; 3 = wgrad9's schedule
    on v_mfma_f32_16x16x32 (wgrad9s, octa_tuning_set(9, 1)); 4 = four waves of 128 x 128 per workgroup (wgrad9a, octa_tuning_set(9, 2))."""
    import ctypes
    from octave_amd import functional as F_
    from octave_amd._lib import WgradJob, lib
    L = lib()
    cases = [(3, 24, 9, 11, 136, 1, 1, 0, 1, True), (2, 40, 13, 10, 130, 3, 1, 1, 1, True), (2, 16, 15, 17, 256, 3, 2, 1, 1, False),
             (2, 32, 12, 12, 256, 3, 1, 1, 2, True), (2, 15, 20, 20, 128, 4, 2, 1, 1, True), (2, 320, 9, 9, 384, 3, 1, 1, 1, False),
             (2, 32, 10, 10, 48, 3, 1, 1, 1, True),
             # 256 x 256 tiles (wgrad9): exact fit, N tail, grouped, strided, 1x1 with a 5-stage pixel axis and a ragged tail
             (2, 256, 10, 12, 256, 3, 1, 1, 1, True), (3, 512, 7, 9, 500, 1, 1, 0, 1, True), (2, 512, 8, 8, 512, 3, 1, 1, 2, True),
             (2, 256, 15, 17, 256, 3, 2, 1, 1, False), (1, 1024, 13, 11, 768, 1, 1, 0, 1, False),
             # enough pixels for an M-split of the 256 x 256 tiles (100 stages of 32 pixels over 9 / 4 tiles)
             (2, 256, 40, 40, 256, 3, 1, 1, 1, True), (3, 512, 36, 30, 512, 1, 1, 0, 2, True)]
    if fold:
        cases = cases * 3         # > 16 jobs with partial tiles: their fold batches must all wait for the kernels that fill them
    gen = torch.Generator().manual_seed(3)
    jobs = (WgradJob * len(cases))()
    keep, want = [], []
    for j, (B, Cin, H, W, Cout, k, s, p, g, bias) in enumerate(cases):
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
        dy = torch.randn(B, Cout, OH, OW, generator=gen).to(dtype).float()
        xd, dyd = F_.to_nhwc(x.to(dev), dtype=dtype, cpad=F_.round8(Cin // g) if g == 1 else Cin), F_.to_nhwc(dy.to(dev), dtype=dtype, cpad=F_.round8(Cout // g) if g == 1 else Cout)
        dw = torch.zeros(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        db = torch.zeros(Cout, device=dev) if bias else None
        d = F_._desc(B, H, W, OH, OW, Cin, Cout, k, k, s, p, g, F_.nhwc_ld(xd), F_.nhwc_ld(dyd), dtype)
        ctypes.memmove(ctypes.byref(jobs[j].d), ctypes.byref(d), ctypes.sizeof(d))
        jobs[j].x, jobs[j].dy, jobs[j].dw, jobs[j].dbias = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr() if bias else None
        for a in range(4):
            jobs[j].dw_strides[a] = dw.stride(a)
        wr = torch.zeros(Cout, Cin // g, k, k, requires_grad=True)
        torch.nn.functional.conv2d(x, wr, None, s, p, 1, g).backward(dy)
        keep.append((xd, dyd, dw, db))
        want.append((wr.grad, dy.sum((0, 2, 3))))
    classes = [int(L.octa_wgrad_job_class(ctypes.byref(jobs[j]))) for j in range(len(cases))]
    assert set(classes) == {0, 1, 2, 3}, classes
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev) if fold else None
    F_.set_wgrad_fold_workspace(ws)
    if fold:
        L.octa_tuning_set(4, 1)       # the batched kernels too (off by default: no gain in situ)
    L.octa_tuning_set(8, sched if sched < 3 else 0)
    L.octa_tuning_set(9, {3: 1, 4: 2}.get(sched, 0))
    try:
        L.octa_conv2d_wgrad_batch(jobs, len(cases), *F_._fold_ws_args(), torch.cuda.current_stream().cuda_stream)
        if sched in (1, 2) and not fold:
            assert "wgrad9x" in L.octa_last_conv_kernel().decode() and ("persistent" in L.octa_last_conv_kernel().decode()) == (sched == 2)
        if sched in (3, 4) and not fold:
            assert ("wgrad9s" if sched == 3 else "wgrad9a") in L.octa_last_conv_kernel().decode()
        first = [(dw.clone(), None if db is None else db.clone()) for _, _, dw, db in keep]
        if fold:
            for _, _, dw, db in keep:
                dw.zero_()
                if db is not None:
                    db.zero_()
            L.octa_conv2d_wgrad_batch(jobs, len(cases), *F_._fold_ws_args(), torch.cuda.current_stream().cuda_stream)
    finally:
        F_.set_wgrad_fold_workspace(None)
        L.octa_tuning_set(4, 0)
        L.octa_tuning_set(8, 0)
        L.octa_tuning_set(9, 0)
    split_jobs = 0
    for j, ((xd, dyd, dw, db), (gw, gb)) in enumerate(zip(keep, want)):
        check(f"wgrad batch job {j} {cases[j]}", dw, gw, 0, 3e-4 * float(gw.abs().max()))
        if db is not None:
            check(f"wgrad batch bias {j}", db, gb, 0, 1e-3 * float(gb.abs().max()) + 1e-3)
        if fold and not torch.equal(first[j][0], dw):
            split_jobs += 1          # a job that kept its float atomics (single-split jobs add straight into dw: order-dependent only across tiles? no: one add per element)
    if fold:
        assert split_jobs == 0, f"{split_jobs} jobs differ between two runs with the fold scratch registered"


# ----------------------------------------------------------------------------------------- round 3: BatchNorm statistics in the conv epilogue
STATS_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, groups, bias, algo            kernel that takes the statistics
    (4, 64, 25, 25, 256, 1, 1, 0, 1, False, 0),       # pointwise, 64x64 tiles (generic 4-wave kernel)
    (3, 256, 13, 13, 72, 1, 1, 0, 1, False, 6),       # N tail (72 channels), 128x64 tiles
    (2, 128, 20, 20, 256, 3, 1, 1, 2, True, 2),       # grouped 3x3 with bias on the 8-wave kernel (256x128 slab)
    (2, 128, 17, 19, 320, 1, 1, 0, 1, False, 3),      # 8-wave 128x256 slab, N tail, odd image (M tail)
    (2, 64, 16, 16, 64, 3, 1, 1, 1, False, 1),        # 3x3 halo kernel: NOT fused, the BatchNorm runs its own pass
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_bn_fused_statistics_vs_separate_pass(dev, case, dtype):
    """layers.conv_bn in 16-bit training mode: the conv sums its own output for the BatchNorm (octa_conv2d_fwd_stats ->
    octa_bn_train_fwd_sums) -- output, batch statistics, running statistics and every gradient against the separate
    statistics pass on the same operands, with a NON-zero running mean as the shift and twice in a row (momentum update)."""
    from octave_amd import functional as F_
    from octave_amd import layers as Ly
    from octave_amd._lib import lib
    B, Cin, H, W, Cout, k, s, p, g, bias, algo = case
    gen = torch.Generator().manual_seed(31)
    x = (torch.randn(B, Cin, H, W, generator=gen) + 0.3).to(dev)
    res = {}
    for fused in (True, False):
        torch.manual_seed(5)
        conv = Ly.Conv2d(Cin, Cout, k, s, p, groups=g, bias=bias).to(dev)
        bn = Ly.BatchNorm2d(Cout).to(dev).train()
        with torch.no_grad():
            bn.running_mean.copy_(torch.linspace(-0.5, 0.5, Cout))
            bn.weight.copy_(torch.linspace(0.5, 1.5, Cout))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, Cout))
        Ly.use_channels_last_weights(conv)
        old, Ly._FUSE_BN_STATS = Ly._FUSE_BN_STATS, fused
        old_min, Ly._FUSE_BN_MIN_BYTES = Ly._FUSE_BN_MIN_BYTES, 0
        F_._ALGO_OVERRIDE = algo
        try:
            xin = F_.to_nhwc(x, dtype=dtype).detach().requires_grad_(True)
            kinds = []
            for _ in range(2):
                y = Ly.conv_bn(conv, bn, xin, relu=True)
                kinds.append(lib().octa_last_conv_kernel().decode())
            gy = torch.randn(tuple(y.shape), generator=torch.Generator().manual_seed(6)).to(dev).to(dtype)
            y.backward(F_.to_nhwc(gy))
        finally:
            Ly._FUSE_BN_STATS = old
            Ly._FUSE_BN_MIN_BYTES = old_min
            F_._ALGO_OVERRIDE = 0
        res[fused] = (F_.to_nchw_f32(y.detach()), bn.running_mean.clone(), bn.running_var.clone(), F_.to_nchw_f32(xin.grad), conv.weight.grad.clone(),
                      bn.weight.grad.clone(), bn.bias.grad.clone(), kinds)
    a, b = res[True], res[False]
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (a[0] - b[0]).abs().max().item() <= 2 * ulp * b[0].abs().max().item() + 1e-6, (case, a[7], (a[0] - b[0]).abs().max().item())
    assert (a[1] - b[1]).abs().max().item() <= 2e-5 * (1.0 + b[1].abs().max().item()), (a[1] - b[1]).abs().max().item()
    assert (a[2] - b[2]).abs().max().item() <= 2e-4 * b[2].abs().max().item(), (a[2] - b[2]).abs().max().item()
    # gradients in relative L2: an output within an ulp of the ReLU threshold may land on the other side (its whole gradient flips
    # on / off), which is a handful of elements, not a statistics error
    for i, nm in ((3, "dx"), (4, "dw"), (5, "dgamma"), (6, "dbeta")):
        rel = (a[i].double() - b[i].double()).norm().item() / max(b[i].double().norm().item(), 1e-30)
        assert rel <= 1e-2, (nm, rel)
    print(f"[conv+bn stats {case}] kernel {a[7][0]}")


# ----------------------------------------------------------------------------------------- round 5: activation derivative in the consumer's data gradient
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["leaky", "sigmoid", "tanh"])
def test_gated_data_gradients_match_dgrad_then_act_bwd(dev, dtype, act):
    """octa_conv2d_dgrad_gated / octa_col2im_taps_gated / octa_fullconv_bwd_gated (functional.ActGate): dx * f'(x) in the consumer's
    epilogue against the unfused pair (data gradient, then octa_act_bwd on the producer's output), and against the closed form in
    float64 -- the discriminator's three consumers: a 1x1 squeeze conv, the 15-channel k4 s2 conv behind torch.cat (only the first
    13 channels are an activation's output), the full-extent head."""
    from octave_amd import functional as F_
    from octave_amd._lib import ACT_LEAKY02, ACT_SIGMOID, ACT_TANH
    code = {"leaky": ACT_LEAKY02, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH}[act]
    gen = torch.Generator().manual_seed(17)

    def act_out(shape):
        z = torch.randn(shape, generator=gen)
        y = {"leaky": torch.nn.functional.leaky_relu(z, 0.2), "sigmoid": torch.sigmoid(z), "tanh": torch.tanh(z)}[act]
        return y.to(dtype).float()          # the values the kernels see

    def fprime(y):
        y = y.double()
        return {"leaky": torch.where(y > 0, 1.0, 0.2), "sigmoid": y * (1 - y), "tanh": 1 - y * y}[act]

    t = TOL[dtype]
    # (a) 1x1 conv 24 -> 13 (its data gradient reaches the 24-channel activation output), 64-channel variant too
    for Cin, Cout, H, W in ((24, 13, 9, 11), (64, 13, 12, 10), (136, 13, 7, 5)):
        x = act_out((2, Cin, H, W))
        w = torch.randn(Cout, Cin, 1, 1, generator=gen) * 0.2
        dy = torch.randn(2, Cout, H, W, generator=gen).to(dtype).float()
        xd = F_.to_nhwc(x.to(dev), dtype=dtype, cpad=F_.round8(Cin))
        dyd = F_.to_nhwc(dy.to(dev), dtype=dtype, cpad=F_.round8(Cout))
        wd = w.to(dev)
        dx_f, applied = F_.raw_conv_dgrad(dyd, wd, (2, Cin, H, W), 1, 0, 1, None, gate=(xd, code, 0))
        assert applied
        dx_u = F_.raw_act_bwd(xd, F_.raw_conv_dgrad(dyd, wd, (2, Cin, H, W), 1, 0, 1), code)
        want = torch.nn.functional.conv_transpose2d(dy.double(), w.double().to(dtype).double() if dtype != torch.float32 else w.double()) * fprime(x)
        scale = float(want.abs().max())
        check(f"gated 1x1 {Cin} vs closed form", dx_f[:, :Cin], want, t["rtol"], t["atol"] * scale)
        check(f"gated 1x1 {Cin} vs unfused", dx_f[:, :Cin], dx_u[:, :Cin], 2 * t["rtol"], 2 * t["atol"] * scale)
    # (b) k4 s2 p1 conv on 15 channels (13 gated + 2 map channels): the tap GEMM + gated fold
    B, Cin, H, W, Cout = 2, 15, 12, 16, 40
    x = torch.cat((act_out((B, 13, H, W)), torch.randn(B, 2, H, W, generator=gen).to(dtype).float()), 1)
    w = torch.randn(Cout, Cin, 4, 4, generator=gen) * 0.1
    OH, OW = H // 2, W // 2
    dy = torch.randn(B, Cout, OH, OW, generator=gen).to(dtype).float()
    xd = F_.to_nhwc(x.to(dev), dtype=dtype, cpad=16)
    dyd = F_.to_nhwc(dy.to(dev), dtype=dtype, cpad=F_.round8(Cout))
    wd = w.to(dev)
    dx_f, applied = F_.raw_conv_dgrad(dyd, wd, (B, Cin, H, W), 2, 1, 1, None, gate=(xd, code, 13))
    assert applied
    wq = w.to(dtype).double() if dtype != torch.float32 else w.double()
    raw = torch.nn.functional.conv_transpose2d(dy.double(), wq, stride=2, padding=1)
    g = torch.ones_like(raw)
    g[:, :13] = fprime(x[:, :13])
    want = raw * g
    scale = float(want.abs().max())
    check("gated k4s2 fold vs closed form", dx_f[:, :Cin], want, 2 * t["rtol"], 2 * t["atol"] * scale)
    # (c) the full-extent head: dx = sign * dout * w * f'(x)
    B, C, H, W = 3, 16, 5, 6
    x = act_out((B, C, H, W))
    wfc = torch.randn(1, C, H, W, generator=gen)
    xd = F_.to_nhwc(x.to(dev), dtype=dtype)
    res = {}
    for gated in (True, False):
        xr = xd.clone().requires_grad_(True)
        wr = wfc.to(dev).requires_grad_(True)
        gate = (F_.ActGate(), code) if gated else None
        out = F_.FullConvFn.apply(xr, wr, None, -1.0, None, gate)
        out.backward(torch.ones_like(out) * 0.5)
        res[gated] = (xr.grad.clone(), wr.grad.clone(), None if gate is None else gate[0].done)
    assert res[True][2] is True
    want = (-0.5 * wfc.double()).expand(B, C, H, W) * fprime(x)
    check("gated head dx vs closed form", res[True][0], want, t["rtol"], t["atol"] * float(want.abs().max()))
    check("gated head dx vs unfused", res[True][0], F_.raw_act_bwd(xd, res[False][0], code), 2 * t["rtol"], 2 * t["atol"] * float(want.abs().max()))
    assert torch.equal(res[True][1], res[False][1])          # the weight gradient does not see the gate


def test_discriminator_backward_without_derivative_kernels(dev):
    """The discriminator's chain with ActGate on: no octa_act_bwd launch in its backward, gradients equal (to bf16 rounding) to the
    chain with the gates off."""
    import importlib
    from octave_amd import functional as F_
    from architectures.discriminator.blocks import DiscriminatorBlock
    torch.manual_seed(3)
    B, H = 2, 64
    net = DiscriminatorBlock(torch.Size((B, 2, H, H)), True, depth=4).to(dev).train()
    net.compute_dtype = torch.bfloat16
    maps = [torch.rand(B, 2, H >> i, H >> i, device=dev, requires_grad=True) for i in range(5)]
    grads = {}
    calls = {}
    orig = F_.raw_act_bwd
    try:
        for on in (True, False):
            n = [0]

            def counting(y, dy, act, _n=n):
                _n[0] += 1
                return orig(y, dy, act)
            F_.raw_act_bwd = counting
            F_._ACT_GATES = on
            st = {k: v.clone() for k, v in net.state_dict().items()}
            torch.manual_seed(11)
            net.zero_grad(set_to_none=True)
            for m in maps:
                m.grad = None
            out = net(maps)
            (out.float() ** 2).sum().backward()
            calls[on] = n[0]
            grads[on] = [m.grad.clone() for m in maps] + [p.grad.clone() for p in net.parameters()]
            net.load_state_dict(st)          # the power iteration moved u / v: same state for the second run
    finally:
        F_.raw_act_bwd = orig
        F_._ACT_GATES = True
    assert calls[True] == 0 and calls[False] == 9, calls
    for i, (a, b) in enumerate(zip(grads[True], grads[False])):
        scale = float(b.abs().max()) + 1e-12
        check(f"discriminator gradient {i}", a, b, 5e-2, 2e-2 * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_discriminator_first_conv_on_the_space_to_depth_image(dev, dtype):
    """The first discriminator conv (k4 s2 p1 on 2 channels) as a k2 s1 conv on NoiseClipS2dFn's output (OCTA_S2D_CONV0, default on)
    against the plain layout: logits, map gradients and every parameter gradient (fp32: 1e-5; bf16: rounding of two summation orders)."""
    from octave_amd import functional as F_
    import architectures.discriminator.blocks as blk
    torch.manual_seed(5)
    B, H = 2, 64
    net = blk.DiscriminatorBlock(torch.Size((B, 2, H, H)), True, depth=4).to(dev).train()
    net.compute_dtype = dtype
    maps = [torch.rand(B, 2, H >> i, H >> i, device=dev, requires_grad=True) for i in range(5)]
    res = {}
    try:
        for on in (True, False):
            blk._S2D_CONV0 = on
            st = {k: v.clone() for k, v in net.state_dict().items()}
            torch.manual_seed(11)
            net.zero_grad(set_to_none=True)
            for m in maps:
                m.grad = None
            out = net(maps)
            (out.float() ** 2).sum().backward()
            res[on] = [out.detach().clone()] + [m.grad.clone() for m in maps] + [p.grad.clone() for p in net.parameters()]
            net.load_state_dict(st)
    finally:
        blk._S2D_CONV0 = True
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        scale = float(b.abs().max()) + 1e-12
        check(f"s2d vs plain, tensor {i}", a, b, tol, tol * scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad2d_patch_kernel_vs_torch(dev, dtype):
    """wgrad2d (octa_tuning_set(10, 1)): the 2-D patch weight-gradient kernel for 3x3 stride-1 layers of exact 5 x 25 geometries, one job per
    launch through octa_conv2d_wgrad_batch, against torch's CPU gradients: one patch, several patches with an M-split, two image
    columns of patches, a grouped layer, an N tail (384 output channels = 1.5 tiles), 96 input channels (3 slices)."""
    import ctypes
    from octave_amd import functional as F_
    from octave_amd._lib import WgradJob, lib
    L = lib()
    cases = [(2, 64, 10, 25, 256, 1), (4, 64, 25, 50, 256, 1), (1, 32, 5, 50, 128, 1), (2, 128, 10, 25, 512, 2), (2, 96, 5, 25, 384, 1), (3, 64, 15, 100, 256, 1)]
    gen = torch.Generator().manual_seed(21)
    L.octa_tuning_set(10, 1)
    keep = []
    try:
        for (B, Cin, H, W, Cout, g) in cases:
            x = torch.randn(B, Cin, H, W, generator=gen).to(dtype).float()
            dy = torch.randn(B, Cout, H, W, generator=gen).to(dtype).float()
            xd, dyd = F_.to_nhwc(x.to(dev), dtype=dtype, cpad=Cin if g > 1 else F_.round8(Cin)), F_.to_nhwc(dy.to(dev), dtype=dtype, cpad=Cout)
            dw = torch.zeros(Cout, Cin // g, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
            d = F_._desc(B, H, W, H, W, Cin, Cout, 3, 3, 1, 1, g, F_.nhwc_ld(xd), F_.nhwc_ld(dyd), dtype)
            jobs = (WgradJob * 1)()
            ctypes.memmove(ctypes.byref(jobs[0].d), ctypes.byref(d), ctypes.sizeof(d))
            jobs[0].x, jobs[0].dy, jobs[0].dw, jobs[0].dbias = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None
            for a in range(4):
                jobs[0].dw_strides[a] = dw.stride(a)
            assert int(L.octa_wgrad_job_class(ctypes.byref(jobs[0]))) == 4
            L.octa_conv2d_wgrad_batch(jobs, 1, None, 0, torch.cuda.current_stream().cuda_stream)
            assert "wgrad2d" in L.octa_last_conv_kernel().decode()
            wr = torch.zeros(Cout, Cin // g, 3, 3, requires_grad=True)
            torch.nn.functional.conv2d(x, wr, None, 1, 1, 1, g).backward(dy)
            check(f"wgrad2d {(B, Cin, H, W, Cout, g)}", dw, wr.grad, 0, 3e-4 * float(wr.grad.abs().max()))
            keep.append((xd, dyd, d, wr.grad))
        # all six layers in ONE call: two launches of the batched kernel (four jobs per launch), a job's blocks starting at a multiple of 8
        jobs = (WgradJob * len(keep))()
        dws = []
        for i, (xd, dyd, d, _) in enumerate(keep):
            dw = torch.zeros_like(keep[i][3]).to(dev).contiguous(memory_format=torch.channels_last)
            ctypes.memmove(ctypes.byref(jobs[i].d), ctypes.byref(d), ctypes.sizeof(d))
            jobs[i].x, jobs[i].dy, jobs[i].dw, jobs[i].dbias = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None
            for a in range(4):
                jobs[i].dw_strides[a] = dw.stride(a)
            dws.append(dw)
        L.octa_conv2d_wgrad_batch(jobs, len(keep), None, 0, torch.cuda.current_stream().cuda_stream)
        for i, (_, _, _, want) in enumerate(keep):
            check(f"wgrad2d batched job {i} {cases[i]}", dws[i], want, 0, 3e-4 * float(want.abs().max()))
    finally:
        L.octa_tuning_set(10, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("use_dice,adv,dev_scale", [(True, True, True), (False, True, False), (True, False, False)])
def test_loss_combine_matches_the_tensor_expression(dev, use_dice, adv, dev_scale):
    """functional.loss_combine (one launch each way) against the sums it replaces, `l[0] + l[1] + kl_w * kl[0] + adv_w * g_adv` times the loss
    scale: bit-identical total, scaled loss and gradients; a NaN in a zero-weight slot (the divergence's NaN flag) must not leak."""
    from octave_amd import functional as F_
    torch.manual_seed(5)
    l = torch.rand(2, device=dev).requires_grad_(True)
    kl2 = torch.tensor([0.37, float("nan")], device=dev).requires_grad_(True) if adv else None
    g_adv = torch.rand((), device=dev).requires_grad_(True) if adv else None
    klw, advw, scale = 0.1, 0.01, 1024.0
    sdev = torch.tensor([scale], device=dev) if dev_scale else None
    total, scaled = F_.loss_combine(l, kl2, g_adv, (1.0, 1.0 if use_dice else 0.0, klw, 0.0, advw), sdev, 1.0 if dev_scale else scale)
    scaled.backward(gradient=F_.one_like_seed(scaled))
    got = [l.grad.clone(), None if kl2 is None else kl2.grad.clone(), None if g_adv is None else g_adv.grad.clone()]
    l2 = l.detach().clone().requires_grad_(True)
    k2 = kl2.detach().clone().requires_grad_(True) if adv else None
    g2 = g_adv.detach().clone().requires_grad_(True) if adv else None
    ref = l2[0] + l2[1] if use_dice else l2[0]
    if adv:
        ref = ref + klw * k2[0] + advw * g2
    ref_scaled = ref * (sdev[0:1].reshape(()) if dev_scale else scale)
    ref_scaled.backward()
    assert not total.requires_grad and torch.equal(total, ref.detach()) and torch.equal(scaled.detach(), ref_scaled.detach())
    assert torch.equal(got[0], l2.grad)
    if adv:
        assert torch.equal(got[1], k2.grad) and torch.equal(got[2], g2.grad) and got[2].shape == g2.grad.shape
