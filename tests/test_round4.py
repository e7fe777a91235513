"""Round-4 reference pins (fixtures written by oracle/gen_golden.py {trainstep400, round4} from the reference itself):
  * the full adversarial step at the HEADLINE resolution, B = 2, 400 x 400 (BASELINE configs[2] geometry: H/16 = 25 is odd ->
    pad / crop of segmentor/compose.py:122-130,142-147; 12 x 12 discriminator head of discriminator/blocks.py:68-72), logits
    included, fp32 + float64 twin;
  * eval-mode ``predict('one-hot')`` at 304 x 304 and 400 x 400 (segmentor/compose.py:189-199) as packed bit masks -- the
    ``dice_vs_ref`` figure of bench.py is the Dice coefficient of the HIP mask against these;
  * fp16 (BASELINE configs[4]: "fp16 with fp32 loss accumulate") against the reference fixtures unet_64.npz / trainstep_48.npz, and
    config 5 at its real size (B = 16, steps alternating 304 x 304 / 400 x 400, static and dynamic loss scale, eager and replay).
CPU half: the oracle against the fixtures; GPU half: the HIP path under the float64-anchored noise-band rule."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BAND, BAND_GRAD
from oracle import ref_ops as R
from oracle.fill import fill_state_dict, hash_input


def _step_inputs(Bn, H):
    x = hash_input((Bn, 1, H, H), 1234).repeat(1, 3, 1, 1)
    u = hash_input((Bn, 1, H, H), 4321)
    ys = torch.zeros(Bn, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    real = F.one_hot((hash_input((Bn, H, H), 999) > 0.8).long(), 2).permute(0, 3, 1, 2).float()
    return x, ys, real


def _bits(G, key, shape):
    n = int(np.prod(shape))
    return np.unpackbits(G[key])[:n].reshape(shape).astype(bool)


# ----------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_trainstep_400(golden):
    """The oracle's full adversarial step at 400 x 400 (B = 2) against the reference's: logits, losses, all gradient norms."""
    from test_oracle import unet_state
    G = golden("trainstep_400.npz")
    Bn, H = 2, 400
    P = unet_state(H, with_disc=True)
    x, ys, real = _step_inputs(Bn, H)
    noise = [torch.from_numpy(G[f"noise{c}"]) for c in range(3)]
    flip = [bool(G[f"uniform{c}"][0] < 0.1) for c in range(3)]
    with torch.no_grad():
        agg = R.resnest_unet_forward(x, {k: v.detach() for k, v in P.items()})[1]
    ref32, ref64 = G["agg"], G["agg_f64_as_f32"].astype(np.float64)
    band = float(np.abs(ref32 - ref64).max())
    # same op sequence as the reference (different thread partition of the sums at most): far inside the reference's own band
    assert float(np.abs(agg.numpy() - ref32).max()) <= 0.05 * band + 1e-4 * float(np.abs(ref32).max())
    l_seg, att, _ = R.segmentor_loss(P, x, ys, noise=noise[0], flip=flip[0])
    l_seg.backward()
    lb = abs(float(G["l_seg"]) - float(G["l_seg_f64"]))
    assert abs(l_seg.item() - float(G["l_seg"])) <= 0.05 * lb + 1e-4 * abs(float(G["l_seg"])), (l_seg.item(), float(G["l_seg"]), lb)
    dev = []
    for k, g in G.items():
        if k.startswith("seg_gradnorm/") and float(g) > 1e-9:
            gn = P["segmentor." + k[13:]].grad.double().norm().item()
            dev.append(abs(gn - float(g)) / float(g))
    assert np.median(dev) <= 2e-3 and np.max(dev) <= 5e-2, (np.median(dev), np.max(dev))
    for p in P.values():
        p.grad = None
    l_d = R.discriminator_loss(P, R.mask_pyramid(real), att, noise[1], flip[1], noise[2], flip[2])
    l_d.backward()
    assert abs(l_d.item() - float(G["l_d"])) <= 1e-3 * abs(float(G["l_d"])) + 1e-6
    for k, g in G.items():
        if k.startswith("disc_gradnorm/"):
            gn = P["discriminator." + k[14:]].grad.double().norm().item()
            assert abs(gn - float(g)) <= 5e-3 * float(g) + 1e-8, (k, gn, float(g))


@pytest.mark.parametrize("H", [304, 400])
def test_oracle_eval_onehot_at_baseline_resolutions(golden, H):
    """Eval-mode forward + predict('one-hot') of the oracle against the reference's packed masks: bit-exact on every pixel whose
    float64 margin is decidable (> 1e-4 x logit scale); the sub-sampled float64 logits within 1e-4 of the logit scale."""
    from test_extras import _eval_state
    G = golden("round4.npz")
    B = 2
    assert G[f"eval{H}/shape"].tolist() == [B, H, H]
    net = _eval_state(H)
    P = {k: v.clone() for k, v in net.state_dict().items()}
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    with torch.no_grad():
        agg = R.resnest_unet_forward(x, P, training=False)[1]
        oh = R.predict(agg, "one-hot").numpy()
    want = _bits(G, f"eval{H}/onehot_cls1_bits", (B, H, H))
    ok = _bits(G, f"eval{H}/decidable_bits", (B, H, H))
    assert ok.mean() > 0.999
    assert np.array_equal(oh[:, 1].astype(bool)[ok], want[ok])
    scale = float(G[f"eval{H}/logit_scale"][0])
    assert float(np.abs(agg[:, :, ::8, ::8].numpy() - G[f"eval{H}/agg_f64_sub8"]).max()) <= 1e-4 * scale


# ----------------------------------------------------------------------------------------- GPU: HIP path
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("these tests need the MI355X (run with -m gpu on the GPU box)")
    return torch.device("cuda:0")


def _norm_band(tag, got, G, pre32, pre64, floor, tail_floor=None, band=BAND_GRAD):
    """Gradient-norm deviations from the float64 reference against the reference's own fp32 deviations (conftest.BAND_GRAD)."""
    tail_floor = 2 * floor if tail_floor is None else tail_floor
    dh, dr = [], []
    top = max(float(g) for k, g in G.items() if k.startswith(pre64))
    for k, g64 in G.items():
        if k.startswith(pre64) and float(g64) > 1e-6 * top and k[len(pre64):] in got:
            name = k[len(pre64):]
            dh.append(abs(got[name] - float(g64)) / float(g64))
            dr.append(abs(float(G[pre32 + name]) - float(g64)) / float(g64))
    assert len(dh) >= 10, (tag, len(dh))
    p95h, p95r = np.percentile(dh, 95), np.percentile(dr, 95)
    print(f"[{tag}] grad-norm deviation from ref64: HIP median {np.median(dh):.2e} p95 {p95h:.2e} max {np.max(dh):.2e}; "
          f"ref32 median {np.median(dr):.2e} p95 {p95r:.2e} max {np.max(dr):.2e}; ratio of medians {np.median(dh) / np.median(dr):.2f}")
    assert np.median(dh) <= band * np.median(dr) + floor and p95h <= band * p95r + tail_floor and np.max(dh) <= 2 * band * np.max(dr) + tail_floor, \
        (tag, np.median(dh), p95h, np.max(dh), np.median(dr), p95r, np.max(dr))


@pytest.mark.gpu
def test_hip_adversarial_step_400_vs_reference(dev, golden):
    """The headline resolution end to end on the HIP path (fp32): B = 2, 400 x 400, logits + four losses + every gradient norm +
    the discriminator step against the reference's step, with the reference's CPU random draws replayed through the global
    generator.  Train-mode band rule: |hip - ref64| <= BAND x |ref32 - ref64| + 1e-4 x scale."""
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    from octave_amd.train import mask_pyramid
    G = golden("trainstep_400.npz")
    Bn, H = 2, 400
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    assert tuple(net.discriminator.out[0].weight.shape[-2:]) == (12, 12)      # blocks.py:68-72 at 400 x 400
    x, ys, real = (t.to(dev) for t in _step_inputs(Bn, H))
    torch.manual_seed(2024)
    att, agg, x4 = net.segmentor(x)
    assert tuple(x4.shape) == (Bn, 2048, 13, 13) and [a.shape[-1] for a in att] == [400, 200, 100, 50, 25]
    ref32, ref64 = G["agg"], G["agg_f64_as_f32"].astype(np.float64)
    noise = float(np.abs(ref32 - ref64).max())
    scale = float(np.abs(ref32).max())
    got = agg.detach().cpu().numpy()
    e64 = float(np.abs(got.astype(np.float64) - ref64).max())
    rms = lambda a: float(np.sqrt((a.astype(np.float64) ** 2).mean()))      # noqa: E731
    e_rms, n_rms = rms(got - ref64), rms(ref32 - ref64)
    print(f"[trainstep 400] logits |hip-ref64| max {e64:.3e} rms {e_rms:.3e}; |ref32-ref64| max {noise:.3e} rms {n_rms:.3e}; ratios {e64 / noise:.2f} / {e_rms / n_rms:.2f}; scale {scale:.1f}")
    # 640 000 logits behind 93 train-mode BatchNorms: the MAXIMUM deviation is one heavy-tailed sample and moves between 1.5x and 3.7x
    # the reference's own from run to run (the float atomics of the split-attention sums re-order; profiles/r04_band_ratios.txt), so the
    # band is held on the RMS deviation and the maximum gets twice the band
    assert e_rms <= BAND * n_rms + 1e-4 * scale, (e_rms, n_rms)
    assert e64 <= 2 * BAND * noise + 1e-4 * scale, (e64, noise)
    margin = np.abs(ref64[:, 0] - ref64[:, 1])
    safe = margin > 10 * noise
    assert safe.sum() >= 1000 and np.array_equal(np.argmax(got, 1)[safe], np.argmax(ref32, 1)[safe])      # (4 % of the pixels have a margin of 10x the reference's own noise here)
    p = torch.softmax(agg, dim=1)
    parts = [net.supervised_loss(p, ys), DiceLoss()(p, ys), InterlayerDivergence()([p, *att]), net.generator_loss(net.discriminator(att))]
    l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
    net.zero_grad()
    l_seg.backward()
    p32, p64 = G["parts"], G["parts_f64"]
    for i, name in enumerate(("wpce", "dice", "kl", "g_adv")):
        print(f"[trainstep 400] {name}: hip {parts[i].item():.6f} ref32 {p32[i]:.6f} ref64 {p64[i]:.6f}")
        assert abs(parts[i].item() - p64[i]) <= BAND * abs(p32[i] - p64[i]) + 2e-4 * abs(p64[i]) + 1e-6, (name, parts[i].item(), p32[i], p64[i])
    l32, l64 = float(G["l_seg"]), float(G["l_seg_f64"])
    assert abs(l_seg.item() - l64) <= BAND * abs(l32 - l64) + 1e-4 * abs(l64), (l_seg.item(), l32, l64)
    gn = {k: q.grad.double().norm().item() for k, q in net.segmentor.named_parameters() if q.grad is not None}
    # The head is downstream of everything chaotic (its gradient comes from WPCE + Dice + KL's P side only): pinned tight.
    for k in ("fc.weight", "fc.bias"):
        g32, g64 = float(G[f"seg_gradnorm/{k}"]), float(G[f"seg_gradnorm_f64/{k}"])
        print(f"[trainstep 400] |grad {k}| hip {gn[k]:.6e} ref32 {g32:.6e} ref64 {g64:.6e}")
        assert abs(gn[k] - g64) <= BAND * abs(g32 - g64) + 1e-3 * g64, (k, gn[k], g32, g64)
    # Everything upstream of the attention maps carries the LS-GAN term's gradient here (g_adv = 180 with the closed-form weights,
    # i.e. ~18 of the loss of ~20), which passes the discriminator's saturated tanh stack: one heavy-tailed shared factor per
    # evaluation.  The reference's own fp32 draw is 1.3 % (median) in this fixture and 15 % in the 304 x 304 one; the HIP path
    # measured 14 % here and 11 % there.  The floors are the reference's own 304 x 304 numbers (median 0.154, p95 0.53, max 1.96).
    _norm_band("trainstep 400 seg", gn, G, "seg_gradnorm/", "seg_gradnorm_f64/", floor=0.154, tail_floor=0.53)
    # Structure: a lost or doubled gradient path puts a parameter AND everything upstream of it off by ~100 % -- dozens of tensors.
    # A single tensor may be: with B = 2 the BatchNorm inside each split-attention block normalises over TWO samples, its 1 / sigma is
    # unbounded, and the block's fc1 / bn1 gradients are the heavy tail of this fixture in the reference's own fp32 run too (its top
    # deviations: encoder_*.conv2.fc1.weight / bn1.* at 7-11 %); the HIP path's LARGEST deviation over eight runs was 27 % .. 92 %, every
    # time one of those tensors (profiles/r04_band_ratios.txt).
    # So: at most 1 % of the tensors beyond 75 %, none beyond 300 %.
    top = max(float(g) for k, g in G.items() if k.startswith("seg_gradnorm_f64/"))
    devs = {k: abs(v - float(G[f"seg_gradnorm_f64/{k}"])) / float(G[f"seg_gradnorm_f64/{k}"]) for k, v in gn.items()
            if float(G[f"seg_gradnorm_f64/{k}"]) > 1e-6 * top}
    far = sorted(((d, k) for k, d in devs.items() if d > 0.75), reverse=True)
    worst = max(devs.items(), key=lambda kv: kv[1])
    print(f"[trainstep 400] largest gradient-norm deviation {worst[1]:.2f} ({worst[0]}); beyond 75 % of the float64 value: {far}")
    assert len(far) <= max(1, len(devs) // 100) and (not far or far[0][0] <= 3.0), far
    net.zero_grad()
    l_d = net.discriminatorial_loss(net.discriminator(mask_pyramid(real)), net.discriminator([a.detach() for a in att]))
    l_d.backward()
    d32, d64 = float(G["l_d"]), float(G["l_d_f64"])
    assert abs(l_d.item() - d64) <= BAND * abs(d32 - d64) + 2e-4 * abs(d64), (l_d.item(), d32, d64)
    gnd = {k: q.grad.double().norm().item() for k, q in net.discriminator.named_parameters()}
    _norm_band("trainstep 400 disc", gnd, G, "disc_gradnorm/", "disc_gradnorm_f64/", floor=2e-3)


def test_oracle_trainstep_400c_conditioned(golden):
    """The oracle's step on the CONDITIONED 400 x 400 fixture (B = 4, discriminator head scaled by COND_SCALE): logits, losses, gradient
    norms and the sixteen full gradients against the reference's."""
    from oracle.fill import COND_SCALE
    from oracle.shapes import octa_state_shapes
    from test_oracle import make_state
    G = golden("trainstep_400c.npz")
    Bn, H = 4, 400
    assert float(G["cond_scale_head"][0]) == COND_SCALE["discriminator.out.0.weight"]
    P = make_state(octa_state_shapes(H, with_disc=True))
    with torch.no_grad():
        P["discriminator.out.0.weight"].mul_(COND_SCALE["discriminator.out.0.weight"])
    x, ys, real = _step_inputs(Bn, H)
    noise = [torch.from_numpy(G[f"noise{c}"]) for c in range(3)]
    flip = [bool(G[f"uniform{c}"][0] < 0.1) for c in range(3)]
    l_seg, att, agg = R.segmentor_loss(P, x, ys, noise=noise[0], flip=flip[0])
    ref32, ref64 = G["agg"], G["agg_f64_as_f32"].astype(np.float64)
    band = float(np.abs(ref32 - ref64).max())
    assert float(np.abs(agg.detach()[:, :, ::2, ::2].numpy() - ref32).max()) <= 0.05 * band + 1e-4 * float(np.abs(ref32).max())
    l_seg.backward()
    assert abs(l_seg.item() - float(G["l_seg"])) <= 1e-5 * abs(float(G["l_seg"]))
    dev_ = [abs(P["segmentor." + k[13:]].grad.double().norm().item() - float(g)) / float(g) for k, g in G.items()
            if k.startswith("seg_gradnorm/") and float(g) > 1e-9]
    assert np.median(dev_) <= 2e-3 and np.max(dev_) <= 5e-2, (np.median(dev_), np.max(dev_))
    from oracle.gen_golden_meta import grad_stride
    for k, g in G.items():
        if k.startswith("seg_grad/"):
            q = P["segmentor." + k[9:]].grad
            got = q.contiguous().flatten()[::grad_stride(q.numel())].numpy()
            g64 = G["seg_grad_f64/" + k[9:]]
            e_ref = np.linalg.norm(g - g64) / np.linalg.norm(g64)
            assert np.linalg.norm(got - g) / np.linalg.norm(g64) <= 0.25 * e_ref + 1e-5, (k, e_ref)


@pytest.fixture
def deterministic():
    """Deterministic mode of the library for one test (octa_tuning_set(5, .): every cross-workgroup sum in a fixed order)."""
    from octave_amd import functional as F_
    F_.set_deterministic(True)
    try:
        yield
    finally:
        F_.set_deterministic(False)


@pytest.mark.gpu
def test_hip_adversarial_step_400c_conditioned_vs_reference(dev, golden, deterministic):
    """The headline resolution on a WELL-CONDITIONED fixture (B = 4: every split-attention bn1 normalises over four samples; the
    discriminator's head scaled so that g_adv ~ 1 instead of 180): the reference's own fp32 gradient norms sit a median of 0.12 % / p95
    0.9 % from its float64 twin here (1.3 % / 5.5 % in trainstep_400), so the bounds can be the ones the review asked for.  The library
    runs in DETERMINISTIC mode: the step is evaluated twice and must be bit-identical (logits, every gradient), and
      * logits: max |hip - ref64| <= BAND x max |ref32 - ref64| (no factor 2), RMS likewise;
      * gradient norms: median <= BAND_GRAD x reference + 0.2 %, p95 <= BAND_GRAD x reference + 1 %, no tensor beyond 25 %;
      * sixteen FULL gradients spread over the ten buckets: relative L2 distance to float64 <= BAND x the reference's own + 0.5 %."""
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    from octave_amd.synth import COND_SCALE
    from octave_amd.train import mask_pyramid
    from oracle.gen_golden_meta import grad_stride
    G = golden("trainstep_400c.npz")
    Bn, H = 4, 400
    x, ys, real = (t.to(dev) for t in _step_inputs(Bn, H))
    runs = []
    for rep in range(2):
        net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
        fill_state_dict(net.state_dict(), scale=COND_SCALE)
        net = net.to(dev).train()
        torch.manual_seed(2024)
        att, agg, x4 = net.segmentor(x)
        p = torch.softmax(agg, dim=1)
        parts = [net.supervised_loss(p, ys), DiceLoss()(p, ys), InterlayerDivergence()([p, *att]), net.generator_loss(net.discriminator(att))]
        l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
        net.zero_grad()
        l_seg.backward()
        grads = {k: q.grad.detach().clone() for k, q in net.segmentor.named_parameters() if q.grad is not None}
        net.zero_grad()
        l_d = net.discriminatorial_loss(net.discriminator(mask_pyramid(real)), net.discriminator([a.detach() for a in att]))
        l_d.backward()
        dgrads = {k: q.grad.detach().clone() for k, q in net.discriminator.named_parameters()}
        runs.append((agg.detach().clone(), [v.item() for v in parts], l_seg.item(), grads, l_d.item(), dgrads))
    # ---- run-to-run: bit-identical in deterministic mode
    a, b = runs
    assert torch.equal(a[0], b[0]), "logits differ between two deterministic runs"
    assert a[1] == b[1] and a[2] == b[2] and a[4] == b[4], (a[1], b[1])
    nd = [k for k in a[3] if not torch.equal(a[3][k], b[3][k])] + [k for k in a[5] if not torch.equal(a[5][k], b[5][k])]
    assert not nd, f"{len(nd)} gradient tensors differ between two deterministic runs: {nd[:8]}"
    agg, parts, l_seg, grads, l_d, dgrads = a
    # ---- logits
    ref32, ref64 = G["agg"], G["agg_f64_as_f32"].astype(np.float64)
    noise, scale = float(np.abs(ref32 - ref64).max()), float(np.abs(ref32).max())
    got = agg[:, :, ::2, ::2].cpu().numpy()
    rms = lambda v: float(np.sqrt((v.astype(np.float64) ** 2).mean()))      # noqa: E731
    e64, e_rms, n_rms = float(np.abs(got.astype(np.float64) - ref64).max()), rms(got - ref64), rms(ref32 - ref64)
    print(f"[trainstep 400c] logits |hip-ref64| max {e64:.3e} rms {e_rms:.3e}; |ref32-ref64| max {noise:.3e} rms {n_rms:.3e}; ratios {e64 / noise:.2f} / {e_rms / n_rms:.2f}; scale {scale:.1f}")
    assert e64 <= BAND * noise + 1e-4 * scale, (e64, noise)
    assert e_rms <= BAND * n_rms + 1e-5 * scale, (e_rms, n_rms)
    margin = np.abs(ref64[:, 0] - ref64[:, 1])
    safe = margin > 10 * noise
    assert safe.mean() > 0.5 and np.array_equal(np.argmax(got, 1)[safe], np.argmax(ref32, 1)[safe])
    # ---- losses
    p32, p64 = G["parts"], G["parts_f64"]
    for i, name in enumerate(("wpce", "dice", "kl", "g_adv")):
        print(f"[trainstep 400c] {name}: hip {parts[i]:.6f} ref32 {p32[i]:.6f} ref64 {p64[i]:.6f}")
        assert abs(parts[i] - p64[i]) <= BAND * abs(p32[i] - p64[i]) + 1e-4 * abs(p64[i]) + 1e-6, (name, parts[i], p32[i], p64[i])
    assert 0.5 <= p64[3] <= 2.0                      # the conditioned LS-GAN term is O(1)
    l32, l64 = float(G["l_seg"]), float(G["l_seg_f64"])
    assert abs(l_seg - l64) <= BAND * abs(l32 - l64) + 1e-4 * abs(l64), (l_seg, l32, l64)
    # ---- gradient norms
    gn = {k: v.double().norm().item() for k, v in grads.items()}
    for k in ("fc.weight", "fc.bias"):
        g32, g64 = float(G[f"seg_gradnorm/{k}"]), float(G[f"seg_gradnorm_f64/{k}"])
        assert abs(gn[k] - g64) <= BAND * abs(g32 - g64) + 1e-3 * g64, (k, gn[k], g32, g64)
    # (measured, deterministic: median 2.4e-3 / p95 8.6e-3 / max 2.7e-2 against the reference's own 1.2e-3 / 9.0e-3 / 5.2e-2:
    # profiles/r05_band_ratios.txt; the review allowed floors of 2 % / 10 %, the measurement supports 0.2 % / 1 %)
    _norm_band("trainstep 400c seg", gn, G, "seg_gradnorm/", "seg_gradnorm_f64/", floor=2e-3, tail_floor=1e-2)
    top = max(float(g) for k, g in G.items() if k.startswith("seg_gradnorm_f64/"))
    devs = {k: abs(v - float(G[f"seg_gradnorm_f64/{k}"])) / float(G[f"seg_gradnorm_f64/{k}"]) for k, v in gn.items()
            if float(G[f"seg_gradnorm_f64/{k}"]) > 1e-6 * top}
    worst = max(devs.items(), key=lambda kv: kv[1])
    print(f"[trainstep 400c] largest gradient-norm deviation {worst[1]:.3f} ({worst[0]})")
    assert worst[1] <= 0.25, worst
    # ---- full gradients
    for k, g32 in G.items():
        if not k.startswith("seg_grad/"):
            continue
        name = k[len("seg_grad/"):]
        g64 = G["seg_grad_f64/" + name]
        q = grads[name]
        got = q.cpu().contiguous().flatten()[::grad_stride(q.numel())].double().numpy()
        assert got.shape == g64.shape, (name, got.shape, g64.shape)
        e_h, e_r = np.linalg.norm(got - g64) / np.linalg.norm(g64), np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
        print(f"[trainstep 400c] grad {name:34s} rel L2 to ref64: hip {e_h:.3e} ref32 {e_r:.3e}")
        assert e_h <= BAND * e_r + 5e-3, (name, e_h, e_r)
    # ---- the discriminator's own step
    d32, d64 = float(G["l_d"]), float(G["l_d_f64"])
    assert abs(l_d - d64) <= BAND * abs(d32 - d64) + 2e-4 * abs(d64), (l_d, d32, d64)
    gnd = {k: v.double().norm().item() for k, v in dgrads.items()}
    _norm_band("trainstep 400c disc", gnd, G, "disc_gradnorm/", "disc_gradnorm_f64/", floor=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("H", [304, 400])
def test_hip_eval_onehot_at_baseline_resolutions(dev, golden, H):
    """predict('one-hot') in eval mode (folded BatchNorm) at both BASELINE resolutions against the reference's mask: bit-exact on
    every decidable pixel, Dice (both classes) = 1 up to the undecidable band, float64 logits within 1e-4 of the logit scale."""
    from octave_amd import functional as F_
    from test_extras import _eval_state
    G = golden("round4.npz")
    B = 2
    net = _eval_state(H).to(dev).eval()
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1).to(dev)
    with torch.no_grad():
        agg = net.segmentor(x)[1]
        oh = net.segmentor.predict(x, "one-hot")[1]
    want = _bits(G, f"eval{H}/onehot_cls1_bits", (B, H, H))
    ok = _bits(G, f"eval{H}/decidable_bits", (B, H, H))
    got = oh[:, 1].cpu().numpy().astype(bool)
    nbad = int((got != want).sum())
    assert np.array_equal(got[ok], want[ok]), f"{int((got != want)[ok].sum())} decidable pixels differ"
    scale = float(G[f"eval{H}/logit_scale"][0])
    e = float(np.abs(agg[:, :, ::8, ::8].cpu().numpy().astype(np.float64) - G[f"eval{H}/agg_f64_sub8"]).max())
    assert e <= 1e-4 * scale, (e, scale)
    ref = torch.from_numpy(np.stack([~want, want], 1).astype(np.float32)).to(dev)
    dice = F_.dice_coefficient(oh.float(), ref)
    print(f"[eval {H}] one-hot vs reference: {nbad} of {got.size} pixels differ (all undecidable), Dice per sample {dice.tolist()}, |hip-ref64| {e:.2e} (scale {scale:.2f})")
    assert float(dice.min()) >= 1 - 1e-4


def _seg_forward(Bn, Hn, dev, dt, x):
    from architectures.models.octa import OctaScribbleNet
    net = OctaScribbleNet(torch.Size((Bn, 3, Hn, Hn)), torch.Size((Bn, 2, Hn, Hn)), True, False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    net.segmentor.compute_dtype = dt
    net.discriminator.compute_dtype = dt
    return net, net.segmentor(x)


@pytest.mark.gpu
def test_unet_fp16_vs_reference(dev, golden):
    """fp16 activations (BASELINE configs[4]) against the reference's 64 x 64 fixture (unet_64.npz).  The train-mode network
    amplifies ANY perturbation chaotically, so the fp16 budget is calibrated by a probe: the HIP fp32 path with
    only the input image rounded to fp16 once.  fp16's distance to the float64 reference may be at most 4x the probe's (plus the
    reference's own fp32 band); the argmax agrees on the high-margin pixels; the WPCE + Dice loss is accumulated in fp32 (the
    reference's 1e-12 epsilons, segmentor/losses.py:38,52, underflow in fp16: the loss kernels take fp32 class maps) and sits
    within the same calibrated band; gradients are finite and the well-conditioned ones next to the output match the reference's."""
    from architectures.segmentor.losses import DiceLoss
    G = golden("unet_64.npz")
    Bn, Hn = 3, 64
    x = hash_input((Bn, 1, Hn, Hn), 1234).repeat(1, 3, 1, 1).to(dev)
    u = hash_input((Bn, 1, Hn, Hn), 4321)
    ys = torch.zeros(Bn, 2, Hn, Hn)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    ys = ys.to(dev)
    ref32, ref64 = G["agg"], G["agg_f64"]
    noise = float(np.abs(ref32 - ref64).max())

    def run(dt, xin):
        net, (att, agg, _) = _seg_forward(Bn, Hn, dev, dt, xin)
        assert agg.dtype == torch.float32 and torch.isfinite(agg).all() and all(torch.isfinite(a).all() for a in att)
        p = torch.softmax(agg, 1)
        loss = net.supervised_loss(p, ys) + DiceLoss()(p, ys)
        assert loss.dtype == torch.float32
        return net, agg, loss
    _, agg_p, loss_p = run(torch.float32, x.half().float())            # the probe: ONE fp16 rounding of the input, fp32 everywhere else
    net, agg, loss = run(torch.float16, x)
    err = lambda a: float(np.sqrt(((a.detach().cpu().numpy().astype(np.float64) - ref64) ** 2).mean()))      # noqa: E731
    e16, ep, ref_rms = err(agg), err(agg_p), float(np.sqrt((ref64 ** 2).mean()))
    print(f"[fp16 unet 64] logits rms error vs ref64: fp16 {e16:.3e}, fp32 with x rounded to fp16 once {ep:.3e}, reference fp32 {float(np.sqrt(((ref32 - ref64) ** 2).mean())):.3e}; "
          f"logit rms {ref_rms:.2f}, max |ref32-ref64| {noise:.3e}")
    assert e16 <= 4 * ep + 2.5 * noise + 0.02 * ref_rms, (e16, ep, noise)
    margin = np.abs(ref64[:, 0] - ref64[:, 1])
    big = margin > 0.25 * np.abs(ref64).max()
    agree = float((torch.argmax(agg, 1).cpu().numpy()[big] == np.argmax(ref64, 1)[big]).mean())
    print(f"[fp16 unet 64] argmax agreement on {int(big.sum())} high-margin pixels: {agree:.4f}")
    assert agree > 0.98
    l64 = float(G["loss_f64"])
    print(f"[fp16 unet 64] loss fp16 {loss.item():.6f}, probe {loss_p.item():.6f}, reference fp32 {float(G['loss']):.6f}, ref64 {l64:.6f}")
    assert abs(loss.item() - l64) <= 4 * abs(loss_p.item() - l64) + 0.02 * abs(l64), (loss.item(), loss_p.item(), l64)
    (loss * 64.0).backward()
    params = dict(net.segmentor.named_parameters())
    for k, q in params.items():
        if q.grad is not None:
            assert torch.isfinite(q.grad).all(), k
    # gradients that do NOT pass through the chaotic part (the head is downstream of everything): against the reference's own
    for k in ("fc.weight", "fc.bias"):
        w = G[f"grad/{k}"]
        g = params[k].grad.cpu().numpy() / 64.0
        rel = float(np.linalg.norm(g - w) / np.linalg.norm(w))
        print(f"[fp16 unet 64] grad {k}: relative L2 error vs reference {rel:.3e}")
        assert rel <= 4 * (ep / ref_rms) + 0.05, (k, rel)


@pytest.mark.gpu
def test_adversarial_step_fp16_vs_reference_48(dev, golden):
    """The 48 x 48 adversarial step of trainstep_48.npz with fp16 activations: the four loss parts (accumulated in fp32 from the
    fp32 class maps) against the reference's float64 values, inside a band calibrated by the probe run (HIP fp32 with the input
    rounded to fp16 once; the LS-GAN term sits behind the discriminator's tanh stack and the chaotic segmentor and moves by tens
    of per cent under ANY perturbation of this closed-form-weight network), every gradient finite, the discriminator step's loss
    within the same band."""
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    from octave_amd.train import mask_pyramid
    G = golden("trainstep_48.npz")
    Bn, H = 6, 48
    x, ys, real = (t.to(dev) for t in _step_inputs(Bn, H))

    def run(dt, xin):
        torch.manual_seed(2024)
        net, (att, agg, _) = _seg_forward(Bn, H, dev, dt, xin)
        p = torch.softmax(agg, dim=1)
        parts = [net.supervised_loss(p, ys), DiceLoss()(p, ys), InterlayerDivergence()([p, *att]), net.generator_loss(net.discriminator(att))]
        assert all(v.dtype == torch.float32 for v in parts)
        return net, att, parts
    _, _, parts_p = run(torch.float32, x.half().float())
    net, att, parts = run(torch.float16, x)
    p64 = G["parts_f64"]
    for i, name in enumerate(("wpce", "dice", "kl", "g_adv")):
        v, vp = parts[i].item(), parts_p[i].item()
        print(f"[fp16 trainstep 48] {name}: fp16 {v:.6f} probe {vp:.6f} ref32 {G['parts'][i]:.6f} ref64 {p64[i]:.6f}")
        if name != "g_adv":
            assert np.isfinite(v) and abs(v - p64[i]) <= 4 * abs(vp - p64[i]) + 0.02 * abs(p64[i]) + 1e-4, (name, v, vp, p64[i])
    # The LS-GAN term sits behind the chaotic segmentor AND the discriminator's tanh stack: whole pixels of the attention maps flip
    # under any perturbation (max |att - att32| ~ 1) and g_adv moves by tens of per cent -- bf16 3.25 -> 6.4, fp32 with the input
    # rounded to bf16 once 4.3, fp16 4.2-4.4 (profiles/r04_disc_fp16_probe.txt).  End to end it is held to a factor of 2; what fp16
    # must get RIGHT is the discriminator's own arithmetic: the same fp32 attention maps through D in fp16 and in fp32.
    assert np.isfinite(parts[3].item()) and 0.5 * p64[3] <= parts[3].item() <= 2.0 * p64[3], (parts[3].item(), p64[3])
    with torch.no_grad():
        att32 = [a.detach().float() for a in att]
        net.discriminator.eval()          # no power iteration: both calls see the same spectral-norm state (the noise is still drawn)
        torch.manual_seed(2024)
        f16 = net.discriminator([a.clone() for a in att32])
        net.discriminator.compute_dtype = torch.float32
        torch.manual_seed(2024)
        f32 = net.discriminator([a.clone() for a in att32])
        net.discriminator.compute_dtype = torch.float16
        net.discriminator.train()
    print(f"[fp16 trainstep 48] D on the same maps: fp16 {f16.flatten().tolist()} fp32 {f32.flatten().tolist()}")
    assert (f16 - f32).abs().max().item() <= 5e-3 * f32.abs().max().item() + 1e-3
    l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
    net.zero_grad()
    (l_seg * 8.0).backward()
    n = 0
    for k, q in net.segmentor.named_parameters():
        if q.grad is not None:
            assert torch.isfinite(q.grad).all(), k
            n += 1
    assert n > 300
    for k in ("fc.weight", "fc.bias"):                 # downstream of the chaotic part: against the reference's float64 norm
        g64 = float(G[f"seg_gradnorm_f64/{k}"])
        gn = dict(net.segmentor.named_parameters())[k].grad.double().norm().item() / 8.0
        print(f"[fp16 trainstep 48] |grad {k}| fp16 {gn:.5e} ref64 {g64:.5e}")
        assert abs(gn - g64) <= 0.10 * g64, (k, gn, g64)
    net.zero_grad()
    l_d = net.discriminatorial_loss(net.discriminator(mask_pyramid(real)), net.discriminator([a.detach() for a in att]))
    l_d.backward()
    d64 = float(G["l_d_f64"])
    print(f"[fp16 trainstep 48] l_d fp16 {l_d.item():.5f} ref64 {d64:.5f}")
    assert np.isfinite(l_d.item()) and abs(l_d.item() - d64) <= 0.5 * abs(d64)          # (its fake branch sees the fp16 attention maps: chaotic, see above)
    for k, q in net.discriminator.named_parameters():
        assert torch.isfinite(q.grad).all(), k


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1024.0, "dynamic"])
def test_train_step_fp16_mixed_resolution_config5_full_size(dev, scale):
    """BASELINE configs[4] at its real per-GPU size: B = 16, steps alternating 304 x 304 and 400 x 400, fp16 activations with the
    losses accumulated in fp32, static (1024) and dynamic loss scale, launched eagerly AND replayed from hipGraphs captured per
    resolution.  Checks: finite losses; per gradient bucket |g_fp16| / |g_fp32| in [0.9, 1.1] against the HIP fp32 step on the same
    weights and inputs; only the discriminator head of the step's resolution moves (blocks.py:68-72: 9 x 9 vs 12 x 12 head)."""
    from architectures.discriminator.blocks import DiscriminatorBlock
    from octave_amd import functional as F_
    from octave_amd.train import TrainStep, mask_pyramid
    from test_train_step import _net
    B, Ha, Hb = 16, 304, 400
    batches = {}
    for H in (Ha, Hb):
        x, ys, real = F_.synth_octa_batch(B, H, H, seed=70 + H, device=dev, vessel=True)
        batches[H] = (x, ys, mask_pyramid(real))

    def build():
        torch.manual_seed(0)
        net = _net(B, Ha, dev, seed_fill=False)
        d_b = DiscriminatorBlock(torch.Size((B, 2, Hb, Hb)), is_training=True, depth=4, num_filters=64).to(dev).train().share_body_with(net.discriminator)
        for d in (net.discriminator, d_b):
            if d._has_noise:
                d.stack_0[0].is_training = False            # same (absent) instance noise in the fp32 and fp16 runs
        return net, d_b
    # fp32 gradients per resolution (lr = 0: the weights stay put, the arenas keep the step's gradients)
    ref = {}
    net, d_b = build()
    st = TrainStep(net, lr=0.0, compute_dtype=torch.float32, extra_discriminators={Hb: d_b})
    try:
        for H in (Ha, Hb):
            torch.manual_seed(3)
            o = st(*batches[H])
            torch.cuda.synchronize()
            ref[H] = ({k: float(v) for k, v in o.items()}, st.seg_arena.g.double().clone(), st.disc_arena.g.double().clone(), list(st.seg_arena.buckets))
    finally:
        st.close()
    del net, d_b, st
    torch.cuda.empty_cache()
    net, d_b = build()
    assert net.discriminator.out[0].weight.shape[-1] == 9 and d_b.out[0].weight.shape[-1] == 12
    st = TrainStep(net, lr=0.0, compute_dtype=torch.float16, loss_scale=scale, extra_discriminators={Hb: d_b})
    try:
        for H in (Ha, Hb):
            for attempt in range(8):                        # a dynamic scale starts at 65536 and may back off first
                torch.manual_seed(3)
                s, n0 = float(st.ls_state[0]), int(st.seg_arena.step_count)
                o = {k: float(v) for k, v in st(*batches[H]).items()}
                torch.cuda.synchronize()
                if int(st.seg_arena.step_count) == n0 + 1:
                    break
                assert scale == "dynamic", f"the static scale {scale} overflowed at {H} x {H}"
            else:
                raise AssertionError("the dynamic loss scale never reached a clean step")
            assert all(np.isfinite(v) for v in o.values()), (H, o)
            o32, g32, d32, bk = ref[H]
            for k in o32:
                assert abs(o[k] - o32[k]) <= 0.02 * abs(o32[k]) + 2e-3, (H, k, o32[k], o[k])
            g16, d16 = st.seg_arena.g.double() / s, st.disc_arena.g.double() / s
            rows = [(tag, (g16[lo:hi].norm() / g32[lo:hi].norm()).item()) for tag, lo, hi in bk]
            rng = st._disc_ranges[id(st._pick_disc(batches[H][0]))]
            dn16 = sum(d16[lo:hi].norm().item() ** 2 for lo, hi in rng) ** 0.5
            dn32 = sum(d32[lo:hi].norm().item() ** 2 for lo, hi in rng) ** 0.5
            rows.append(("discriminator", dn16 / dn32))
            print(f"[config5 {H} scale {scale}] loss scale {s:g}; |g_fp16|/|g_fp32| per bucket: " + "; ".join(f"{t}: {r:.4f}" for t, r in rows))
            for tag, r in rows:
                assert 0.9 <= r <= 1.1, (H, tag, r)
    finally:
        st.close()
    del st
    torch.cuda.empty_cache()
    # training steps (lr > 0): only the head of the step's resolution moves; eager, then captured per resolution and replayed
    st = TrainStep(net, lr=1e-4, compute_dtype=torch.float16, loss_scale=scale, extra_discriminators={Hb: d_b})
    try:
        head = {Ha: net.discriminator.out[0].weight, Hb: d_b.out[0].weight}
        body = net.discriminator.squeeze_dict["squeeze_0"][0].weight

        def one(H):
            other = Hb if H == Ha else Ha
            h0, o0, b0 = head[H].detach().clone(), head[other].detach().clone(), body.detach().clone()
            skipped0 = int(st.disc_arena.step_count)
            out = {k: float(v) for k, v in st(*batches[H]).items()}
            torch.cuda.synchronize()
            assert all(np.isfinite(v) for v in out.values()), (H, out)
            assert torch.equal(head[other], o0), f"the {other} x {other} head moved in a {H} x {H} step"
            if int(st.disc_arena.step_count) > skipped0:             # (a dynamic scale may skip its first updates)
                assert not torch.equal(head[H], h0) and not torch.equal(body, b0)
            return out
        for H in (Ha, Hb, Ha, Hb):
            one(H)
        if scale == "dynamic":
            for _ in range(12):                                     # let the scale settle below the overflow threshold
                one(Ha); one(Hb)
        for H in (Ha, Hb):
            st.capture(*batches[H])
        assert sorted(st._caps) == [Ha, Hb]
        st.launch = "graph"
        n0 = int(st.seg_arena.step_count)
        for H in (Ha, Hb, Ha, Hb):
            one(H)
        assert int(st.seg_arena.step_count) == n0 + 4, "a replayed fp16 step was skipped for a non-finite gradient"
        for k, q in net.segmentor.named_parameters():
            assert torch.isfinite(q).all(), k
        for d in (net.discriminator, d_b):
            for k, q in d.named_parameters():
                assert torch.isfinite(q).all(), k
    finally:
        st.close()
