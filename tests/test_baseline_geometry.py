"""GPU: the dominant conv kernels at the benchmark's REAL launch geometry (BASELINE configs[2]: B = 16, 400 x 400, bf16).

The 8-wave kernels (conv_igemm8 forward / data gradient, wgrad8 batched weight gradient) only run on layers with >= 128
channels per group; the small-shape tests of test_hip_ops.py never reach their M-splits, multi-job batches, magic-number
divisions and 24-bit pixel arithmetic at M = 160 000.  Here every big layer of the U-Net is checked at B = 16 against float64
contractions evaluated on the GPU with torch.matmul on the SAME bf16-rounded operands (a CPU oracle needs minutes at this size):
  * forward and data gradient on a pixel subsample (corners, edges, random interior pixels),
  * weight gradient tap by tap over ALL pixels,
  * the three adjoint identities,
and whole stages are run twice -- deferred + batched weight gradients (wgrad8, the flush the train step performs) against
immediate per-layer launches (the 4-wave kernels) -- and one full TrainStep at B = 16 in bf16 is held against the HIP fp32 step.
Reference layers: extra/resnest.py:24 (decoder 3x3), :50 (ConvTranspose k2 s2), :83-84 (split-attention grouped 3x3)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need the MI355X (run with -m gpu on the GPU box)")
    return torch.device("cuda:0")


BIG_LAYERS = [
    # Cin, Cout, k, s, p, g, B, H, W
    (512, 256, 3, 1, 1, 1, 16, 100, 100),     # decoder_2 3x3            M = 160 000, K = 4608
    (1024, 512, 3, 1, 1, 1, 16, 50, 50),      # decoder_3 3x3            M = 40 000,  K = 9216
    (2048, 1024, 3, 1, 1, 1, 16, 25, 25),     # decoder_4 3x3            M = 10 000,  K = 18 432 (odd image)
    (256, 512, 3, 1, 1, 4, 16, 100, 100),     # decoder_2 split-attention conv, groups 4 (64 -> 128 per group)
    (512, 1024, 1, 1, 0, 1, 16, 50, 50),      # encoder_3 conv3-like pointwise (bottleneck 1x1)
    (512, 1024, 3, 1, 1, 2, 16, 13, 13),      # encoder_4 split-attention conv, groups 2: 88 tiles on 256 CUs (tail split with groups)
    (1024, 2048, 3, 1, 1, 4, 16, 25, 25),     # decoder_4 split-attention conv, groups 4: 632 tiles
]


def _pixels(B, H, W, n, gen):
    """(b, h, w) index triples: the four corners and edge midpoints of two images + random interior pixels."""
    pts = [(b, h, w) for b in (0, B - 1) for h in (0, H // 2, H - 1) for w in (0, W // 2, W - 1)]
    rb = torch.randint(0, B, (n,), generator=gen).tolist()
    rh = torch.randint(0, H, (n,), generator=gen).tolist()
    rw = torch.randint(0, W, (n,), generator=gen).tolist()
    pts += list(zip(rb, rh, rw))
    t = torch.tensor(pts, dtype=torch.long)
    return t[:, 0], t[:, 1], t[:, 2]


def _ref_fwd_pixels(x64, w64, b, oh, ow, s, p, g):
    """y[b, :, oh, ow] in float64 for the selected output pixels; x64 (B,Cin,H,W), w64 (Cout,Cin/g,k,k) on the GPU."""
    Cout, Cg, k, _ = w64.shape
    xp = torch.nn.functional.pad(x64, (p, p, p, p))
    out = torch.zeros((b.numel(), Cout), dtype=torch.float64, device=x64.device)
    cog = Cout // g
    for kh in range(k):
        for kw in range(k):
            xs = xp[b, :, oh * s + kh, ow * s + kw]                       # (P, Cin)
            for gi in range(g):
                out[:, gi * cog:(gi + 1) * cog] += xs[:, gi * Cg:(gi + 1) * Cg] @ w64[gi * cog:(gi + 1) * cog, :, kh, kw].t()
    return out


def _ref_dgrad_pixels(dy64, w64, b, ih, iw, s, p, g, Cin):
    """dx[b, :, ih, iw] in float64 for the selected input pixels."""
    Cout, Cg, k, _ = w64.shape
    B, _, OH, OW = dy64.shape
    out = torch.zeros((b.numel(), Cin), dtype=torch.float64, device=dy64.device)
    cog = Cout // g
    for kh in range(k):
        for kw in range(k):
            th, tw = ih + p - kh, iw + p - kw
            ok = (th >= 0) & (tw >= 0) & (th % s == 0) & (tw % s == 0) & (th // s < OH) & (tw // s < OW)
            oh, ow = (th // s).clamp(0, OH - 1), (tw // s).clamp(0, OW - 1)
            ds = dy64[b, :, oh, ow] * ok[:, None].to(torch.float64)       # (P, Cout)
            for gi in range(g):
                out[:, gi * Cg:(gi + 1) * Cg] += ds[:, gi * cog:(gi + 1) * cog] @ w64[gi * cog:(gi + 1) * cog, :, kh, kw]
    return out


def _ref_wgrad_tap(x64, dy64, kh, kw, s, p, g):
    """dw[:, :, kh, kw] in float64 over ALL pixels: (Cout, Cin/g)."""
    B, Cin, H, W = x64.shape
    _, Cout, OH, OW = dy64.shape
    xp = torch.nn.functional.pad(x64, (p, p, p, p))
    xs = xp[:, :, kh:kh + (OH - 1) * s + 1:s, kw:kw + (OW - 1) * s + 1:s].permute(0, 2, 3, 1).reshape(-1, Cin)
    dm = dy64.permute(0, 2, 3, 1).reshape(-1, Cout)
    Cg, cog = Cin // g, Cout // g
    return torch.cat([dm[:, gi * cog:(gi + 1) * cog].t() @ xs[:, gi * Cg:(gi + 1) * Cg] for gi in range(g)], dim=0)


@pytest.mark.parametrize("case", BIG_LAYERS)
def test_big_layer_kernels_at_baseline_geometry(dev, case):
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    Cin, Cout, k, s, p, g, B, H, W = case
    L = lib()
    gen = torch.Generator(device="cpu").manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * (1.0 / (k * k * Cin / g) ** 0.5)).to(dev)
    wq = torch.nn.Parameter(w.bfloat16().float().contiguous(memory_format=torch.channels_last))
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, Cout, OH, OW, generator=gen).to(dev).to(torch.bfloat16)
    xn, dyn = F_.to_nhwc(x), F_.to_nhwc(dy)
    x64, dy64, w64 = x.double(), dy.double(), wq.detach().double()
    pb, ph, pw = [t.to(dev) for t in _pixels(B, OH, OW, 160, gen)]
    ref_y = _ref_fwd_pixels(x64, w64, pb, ph, pw, s, p, g)
    ib, ih, iw = [t.to(dev) for t in _pixels(B, H, W, 160, gen)]
    ref_dx = _ref_dgrad_pixels(dy64, w64, ib, ih, iw, s, p, g, Cin)
    seen = set()
    sk_ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    for algo in (0, 2, 3, 8, 1, 102, 103, 12, 112, 9, 10, 7):      # 12 / 112: the 2-D patch kernel (halo8.hpp) without / with the scratch; 9 / 10: pwgemm.hpp (1x1 layers); 7: resident weights
        # 102 / 103: the 8-wave kernel with the tail-split scratch registered (octa_conv_desc.ws): the 316-tile (25 x 25)
        # and 626-tile (50 x 50) launches then run their last 60 / 114 tiles as 4 / 2 workgroups each + the fix-up launch
        split = algo >= 100
        F_._ALGO_OVERRIDE = algo % 100
        if split:
            F_.set_splitk_workspace(sk_ws)
        try:
            y = F_.raw_conv_fwd(xn, wq, None, s, p, g)
            kf = L.octa_last_conv_kernel().decode()
            dx = F_.raw_conv_dgrad(dyn, wq, (B, Cin, H, W), s, p, g)
            kd = L.octa_last_conv_kernel().decode()
        finally:
            F_._ALGO_OVERRIDE = 0
            F_.set_splitk_workspace(None)
        if algo == 112:
            pass                                                     # (halo8 splits where ITS tile count leaves a short last round)
        elif split and k == 3 and ((g == 1 and H <= 50) or (g > 1 and H <= 25)):
            assert "+tail" in kf or "+tail" in kd, (kf, kd)          # (the data gradient's N is Cin: its tile count differs)
        elif not split:
            assert "+tail" not in kf and "+tail" not in kd, (kf, kd)
        algo = algo % 100
        seen.update((kf.split("<")[0], kd.split("<")[0]))
        got_y = y[pb, :, ph, pw].double()
        err = (got_y - ref_y).abs()
        assert bool((err <= 2.0 ** -8 * ref_y.abs() + 2e-3 * ref_y.abs().max()).all()), (case, algo, kf, err.max().item(), ref_y.abs().max().item())
        got_dx = dx[ib, :, ih, iw].double()
        err = (got_dx - ref_dx).abs()
        assert bool((err <= 2.0 ** -8 * ref_dx.abs() + 2e-3 * ref_dx.abs().max()).all()), (case, algo, kd, err.max().item(), ref_dx.abs().max().item())
        if algo in (2, 3):
            assert "conv_igemm8_kernel" in kf and "conv_igemm8_kernel" in kd, (kf, kd)
        if algo == 12 and k == 3 and s == 1:
            assert "conv_halo8_kernel" in kf and "conv_halo8_kernel" in kd, (kf, kd)
        if algo in (9, 10) and k == 1:
            assert "pwgemm_kernel" in kf and "pwgemm_kernel" in kd, (kf, kd)
        if algo == 7 and k == 3 and Cin // g == 64:
            assert "conv_res3x3" in kf, kf                           # 64 -> 128 per group: two 64-channel output slices per group, compact stages
        # adjoint identity fwd <-> dgrad over the WHOLE tensors (any wrong tile anywhere breaks it)
        t_f = (F_.to_nchw_f32(y).double() * dy64).sum().item()
        t_d = (F_.to_nchw_f32(dx).double() * x64).sum().item()
        scale = (F_.to_nchw_f32(y).double().abs() * dy64.abs()).sum().item()
        assert abs(t_f - t_d) <= 2e-4 * scale, (case, algo, t_f, t_d, scale)
    # weight gradient: the batched 8-wave kernel (what a TrainStep flush launches) and the immediate per-layer kernel
    taps = sorted({(0, 0), (k // 2, k // 2), (k - 1, k // 2)})
    refs = {t: _ref_wgrad_tap(x64, dy64, t[0], t[1], s, p, g) for t in taps}
    # (the ungrouped 3x3 layers with >= 256 channels on both sides leave the batch for the 2-D patch kernel, wgrad2d: both kernels are checked there)
    for mode in ("batched", "batched-no-wgrad2d", "immediate"):
        dw = torch.zeros(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        if mode.startswith("batched"):
            F_.defer_wgrads(True)
            L.octa_tuning_set(10, 2 if mode == "batched" else 0)
            try:
                F_.raw_conv_wgrad(xn, dyn, wq, s, p, g, dw=dw, defer=True)
                assert F_.pending_wgrads() == 1
                job = F_._WGRAD_Q[0][0]
                cls = int(L.octa_wgrad_job_class(ctypes.byref(job)))
                patch = mode == "batched" and k == 3 and s == 1 and g == 1 and Cin >= 256 and Cout >= 256 and H % 5 == 0 and W % 25 == 0
                assert cls == 4 if patch else cls in (1, 2, 3), (cls, "this layer must run on the batched 8-wave kernels" if not patch else "... on wgrad2d")
                F_.flush_wgrads()
            finally:
                F_.defer_wgrads(False)
                L.octa_tuning_set(10, 2)
            assert any(k in L.octa_last_conv_kernel().decode() for k in (("wgrad2d",) if patch else ("wgrad8", "wgrad9"))), L.octa_last_conv_kernel().decode()
        else:
            F_.raw_conv_wgrad(xn, dyn, wq, s, p, g, dw=dw)
        for (kh, kw), ref in refs.items():
            err = (dw[:, :, kh, kw].double() - ref).abs().max().item()
            assert err <= 3e-4 * ref.abs().max().item(), (case, mode, (kh, kw), err, ref.abs().max().item())
        t_w = (dw.double() * w64).sum().item()
        assert abs(t_w - t_f) <= 2e-4 * scale, (case, mode, t_w, t_f, scale)
    print(f"[baseline geometry {case}] kernels: {sorted(seen)}")


def test_upshuffle_layer_at_baseline_geometry(dev):
    """Upsampling 64 -> 64 (ConvTranspose2d k2 s2 + bias, extra/resnest.py:50) at 16 x 64 x 200 x 200 -> 400 x 400: forward,
    data gradient and weight / bias gradients against FULL float64 contractions (four pointwise GEMMs)."""
    from octave_amd import functional as F_
    B, Cin, Cout, H, W = 16, 64, 64, 200, 200
    gen = torch.Generator(device="cpu").manual_seed(23)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev).to(torch.bfloat16)
    w = torch.nn.Parameter((torch.randn(Cin, Cout, 2, 2, generator=gen) * 0.125).bfloat16().float().to(dev).contiguous(memory_format=torch.channels_last))
    bias = torch.nn.Parameter(torch.randn(Cout, generator=gen).to(dev))
    xin = F_.to_nhwc(x).requires_grad_(True)
    y = F_.conv_transpose2x2(xin, w, bias)
    dy = torch.randn(B, Cout, 2 * H, 2 * W, generator=gen).to(dev).to(torch.bfloat16)
    y.backward(F_.to_nhwc(dy))
    xm = x.double().permute(0, 2, 3, 1).reshape(-1, Cin)
    yd, dyd = F_.to_nchw_f32(y.detach()).double(), dy.double()
    dx_ref = torch.zeros_like(xm)
    for di in range(2):
        for dj in range(2):
            wt = w.detach().double()[:, :, di, dj]                                       # (Cin, Cout)
            ref = (xm @ wt + bias.detach().double()).reshape(B, H, W, Cout).permute(0, 3, 1, 2)
            got = yd[:, :, di::2, dj::2]
            assert bool(((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 2e-3).all()), (di, dj, (got - ref).abs().max().item())
            dym = dyd[:, :, di::2, dj::2].permute(0, 2, 3, 1).reshape(-1, Cout)
            dx_ref += dym @ wt.t()
            dw_ref = xm.t() @ dym
            err = (w.grad[:, :, di, dj].double() - dw_ref).abs().max().item()
            assert err <= 3e-4 * dw_ref.abs().max().item(), (di, dj, err)
    got_dx = F_.to_nchw_f32(xin.grad).double().permute(0, 2, 3, 1).reshape(-1, Cin)
    assert bool(((got_dx - dx_ref).abs() <= 2.0 ** -8 * dx_ref.abs() + 2e-3 * dx_ref.abs().max()).all())
    db_ref = dyd.sum(dim=(0, 2, 3))
    assert (bias.grad.double() - db_ref).abs().max().item() <= 1e-4 * (B * 4 * H * W) ** 0.5 + 1e-3 * db_ref.abs().max().item()


STAGES = [
    # (module path on ResnestUNet, input shape)
    ("encoder_3", (16, 512, 50, 50)),       # six bottlenecks at 25 x 25: the 19-job encoder_3 flush of the train step
    ("decoder_3", (16, 1024, 50, 50)),      # ResNestDecoder 1024 -> 512: 3x3, shortcut 1x1, grouped split-attention conv
    ("decoder_2", (16, 512, 100, 100)),     # ResNestDecoder 512 -> 256 at 100 x 100 (M = 160 000)
    ("upsampling_3", (16, 1024, 25, 25)),   # ConvTranspose 1024 -> 512
]


@pytest.mark.parametrize("stage", STAGES)
def test_deferred_batched_wgrads_equal_immediate_at_baseline_geometry(dev, stage):
    """A whole stage at B = 16: the backward pass queues its weight-gradient jobs (what a TrainStep does); the SAME jobs -- same
    activation and gradient buffers -- are then run twice: flushed as batched 8-wave launches (multi-job batches, M-split chosen
    over all jobs) and one by one on the per-layer kernels.  (Two separate backward passes are not comparable this tightly: the
    split-attention and BatchNorm reductions use float atomics and the 16-sample bn1 amplifies their rounding noise, so the
    incoming gradients differ from run to run.)"""
    from architectures.segmentor.compose import ResnestUNet
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    L = lib()
    name, shape = stage
    torch.manual_seed(0)
    unet = ResnestUNet(2, False).to(dev).train()
    mod = getattr(unet, name)
    params = [(n, p) for n, p in mod.named_parameters()]
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = F_.to_nhwc(torch.randn(*shape, generator=gen).to(dev).to(torch.bfloat16))
    for _, p in params:
        p.grad = torch.zeros_like(p.data)                  # same strides as the (channels-last) parameter: the gradient sink
    F_.set_grad_sink(True)
    F_.defer_wgrads(True)
    try:
        xin = x.detach().clone().requires_grad_(True)
        y = mod(xin)
        y = y[0] if isinstance(y, tuple) else y
        g = torch.randn(tuple(y.shape), generator=torch.Generator(device="cpu").manual_seed(9)).to(dev).to(torch.bfloat16)
        y.backward(F_.to_nhwc(g))
        jobs = list(F_._WGRAD_Q)
        classes = [int(L.octa_wgrad_job_class(ctypes.byref(j))) for j, _ in jobs]
        assert len(jobs) >= 1 and any(c > 0 for c in classes), classes
        targets = {}
        for j, keep in jobs:
            targets[j.dw] = keep[4]
            if keep[5] is not None:
                targets[j.dbias] = keep[5]
        before = {k: t.detach().clone() for k, t in targets.items()}        # what the non-deferred part of backward already added
        F_.flush_wgrads()
        torch.cuda.synchronize()
        batched = {k: t.detach().clone() for k, t in targets.items()}
        for k, t in targets.items():
            t.copy_(before[k])
        st = torch.cuda.current_stream().cuda_stream
        for j, keep in jobs:
            L.octa_conv2d_wgrad(ctypes.byref(j.d), j.x, j.dy, j.dw, j.dw_strides, j.dbias, st)
        torch.cuda.synchronize()
    finally:
        F_.defer_wgrads(False)
        F_.set_grad_sink(False)
    worst = 0.0
    for k, t in targets.items():
        a, b = (batched[k] - before[k]).double(), (t.detach() - before[k]).double()
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        den = b.norm().item()
        assert den > 0
        rel = (a - b).norm().item() / den
        worst = max(worst, rel)
        # both are fp32 accumulations of the same bf16 products; only the summation order differs.  A bias gradient is a plain
        # column sum over 10^4 .. 10^5 pixels whose terms cancel (|sum| << sum |terms|), hence the wider band for 1-D targets
        assert rel <= (2e-3 if t.dim() == 1 else 2e-4), (name, tuple(t.shape), rel, den)
    print(f"[stage {name}] {len(jobs)} queued jobs, classes {sorted(set(classes))}; worst relative L2 difference batched vs per-layer: {worst:.2e}")


@pytest.mark.parametrize("cfg", [(400, True), (304, False)])
def test_full_train_step_b16_bf16_vs_fp32(dev, cfg):
    """BASELINE configs[2] (B = 16, 400 x 400, full adversarial step) and configs[1] (304 x 304, segmentor-only) as ONE TrainStep
    launched eagerly in bf16, against the HIP fp32 step on the same weights and inputs, per gradient BUCKET (the all-reduce unit,
    in gradient-completion order): losses within 2 %, |g_bf16| / |g_fp32| within [0.90, 1.10], and the DIRECTION held to a
    self-calibrated band.  The gradient of this network is ill-conditioned beyond the decoder (train-mode BatchNorm; the
    split-attention bn1 normalises over the 16 samples of the batch; ReLU masks flip): a THIRD run -- fp32 everywhere, only the
    input image rounded to bf16 once -- already decorrelates the encoder buckets (measured, profiles/r03_bf16_conditioning.txt:
    cosine to the clean fp32 gradient 0.9998 / 0.965 / 0.85 / 0.58 / 0.51 from the decoder to the stem at 400 x 400; two clean fp32
    runs agree to 0.9997).  So: 1 - cos(bf16) <= 4 (1 - cos(one rounding)) + 0.02 per bucket, and >= 0.98 for the first
    (decoder-side) bucket outright."""
    from octave_amd import functional as F_
    from octave_amd.train import TrainStep, mask_pyramid
    from test_train_step import _net
    H, adv = cfg
    B = 16
    x, ys, real = F_.synth_octa_batch(B, H, H, seed=77, device=dev, vessel=True)
    pyr = mask_pyramid(real)
    out, grads, bk = {}, {}, None
    for tag, dt, xin in (("fp32", torch.float32, x), ("fp32, x rounded", torch.float32, x.bfloat16().float()), ("bf16", torch.bfloat16, x)):
        torch.manual_seed(0)
        net = _net(B, H, dev, seed_fill=False)
        if adv and net.discriminator._has_noise:
            net.discriminator.stack_0[0].is_training = False        # same (absent) instance noise in every run
        st = TrainStep(net, lr=0.0, compute_dtype=dt, adversarial=adv)
        try:
            torch.manual_seed(3)
            o = st(xin, ys, pyr if adv else None)
            torch.cuda.synchronize()
            out[tag] = {k: float(v) for k, v in o.items()}
            grads[tag] = (st.seg_arena.g.double().clone(), st.disc_arena.g.double().clone() if adv else None)
            bk = list(st.seg_arena.buckets)
        finally:
            st.close()
        del net, st
        torch.cuda.empty_cache()
    o32, o16 = out["fp32"], out["bf16"]
    for k in o32:
        assert np.isfinite(o16[k]) and abs(o16[k] - o32[k]) <= 0.02 * abs(o32[k]) + 2e-3, (k, o32[k], o16[k])

    def stats(a, b):
        return b.norm().item() / a.norm().item(), (a @ b).item() / (a.norm().item() * b.norm().item())
    rows = []
    for tag, lo, hi in bk:
        ref = grads["fp32"][0][lo:hi]
        rows.append((tag, *stats(ref, grads["bf16"][0][lo:hi]), stats(ref, grads["fp32, x rounded"][0][lo:hi])[1]))
    if adv:
        ref = grads["fp32"][1]
        rows.append(("discriminator", *stats(ref, grads["bf16"][1]), stats(ref, grads["fp32, x rounded"][1])[1]))
    print(f"[B16 {H} {'adversarial' if adv else 'seg-only'}] losses fp32 {o32} bf16 {o16}")
    print("    bucket: |g_bf16|/|g_fp32|, cos(bf16, fp32), cos(fp32 with x rounded once, fp32): " + "; ".join(f"{t}: {r:.4f}, {c:.4f}, {cp:.4f}" for t, r, c, cp in rows))
    for i, (tag, ratio, cos, cos_pert) in enumerate(rows):
        assert 0.90 <= ratio <= 1.10, (tag, ratio)
        assert 1.0 - cos <= 4.0 * (1.0 - cos_pert) + 0.02, (tag, cos, cos_pert)
        if i == 0 or tag == "discriminator":
            assert cos >= 0.98, (tag, cos)
