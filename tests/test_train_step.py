"""GPU: the full adversarial training step on the HIP path against the golden
vectors of the reference (tests/golden/trainstep_48.npz) and against torch.optim.Adam."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.fill import fill_state_dict, hash_input

pytestmark = pytest.mark.gpu


def _inputs(Bn, H, dev):
    x = hash_input((Bn, 1, H, H), 1234).repeat(1, 3, 1, 1).to(dev)
    u = hash_input((Bn, 1, H, H), 4321)
    ys = torch.zeros(Bn, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    real = F.one_hot((hash_input((Bn, H, H), 999) > 0.8).long(), 2).permute(0, 3, 1, 2).float()
    return x, ys.to(dev), real.to(dev)


def test_adversarial_step_losses_and_grads_vs_reference(golden):
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    from octave_amd.train import mask_pyramid
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    G = golden("trainstep_48.npz")
    Bn, H = 6, 48
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    x, ys, real = _inputs(Bn, H, dev)
    from oracle import ref_ops as R
    P = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    torch.manual_seed(2024)        # the three discriminator calls consume (normal, uniform) x 3 from the CPU generator
    att, agg, _ = net.segmentor(x)
    p = torch.softmax(agg, dim=1)
    kl = InterlayerDivergence()
    kl.check_nan = False
    parts = [net.supervised_loss(p, ys), DiceLoss()(p, ys), kl([p, *att]), net.generator_loss(net.discriminator(att))]
    l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
    # (a) stage-wise: the oracle's losses / discriminator evaluated on the HIP path's own (att, agg)
    att_c, p_c, ys_c = [a.detach().cpu() for a in att], p.detach().cpu(), ys.cpu()
    noise = [torch.from_numpy(G[f"noise{c}"]) for c in range(3)]
    flip = [bool(G[f"uniform{c}"][0] < 0.1) for c in range(3)]
    with torch.no_grad():
        want = [R.weighted_partial_ce(p_c, ys_c, 2), R.dice_loss(p_c, ys_c), R.interlayer_divergence([p_c, *att_c]),
                R.ls_generator_loss(R.discriminator_forward(att_c, {k: v.clone() for k, v in P.items()}, noise=noise[0], flip=flip[0]))]
    for name, got, w in zip(("wpce", "dice", "kl", "g_adv"), parts, want):
        assert abs(got.item() - w.item()) <= 1e-4 * abs(w.item()) + 1e-6, (name, got.item(), w.item())
    net.zero_grad()
    l_seg.backward()
    # (b) end to end against the reference's step: within 4x the band the reference's own fp32 result
    # keeps around its float64 evaluation (DESIGN.md "Parity and conditioning")
    l64, l32 = float(G["l_seg_f64"]), float(G["l_seg"])
    assert abs(l_seg.item() - l64) <= 4 * abs(l32 - l64) + 1e-4 * abs(l64), (l_seg.item(), l32, l64)
    params = dict(net.segmentor.named_parameters())
    dh, dr = [], []
    for k, g in G.items():
        if k.startswith("seg_gradnorm_f64/") and float(g) > 1e-9:
            name = k[len("seg_gradnorm_f64/"):]
            gn = params[name].grad.double().norm().item()
            dh.append(abs(gn - float(g)) / float(g))
            dr.append(abs(float(G["seg_gradnorm/" + name]) - float(g)) / float(g))
    print(f"[trainstep] l_seg {l_seg.item():.6f} (ref32 {l32:.6f}, ref64 {l64:.6f}); grad-norm deviation from ref64: HIP median {np.median(dh):.2e} "
          f"max {np.max(dh):.2e}; ref32 median {np.median(dr):.2e} max {np.max(dr):.2e}")
    assert np.median(dh) <= 4 * np.median(dr) + 1e-3 and np.max(dh) <= 4 * np.max(dr) + 2e-3
    net.zero_grad()
    real_pyr = mask_pyramid(real)
    P2 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}     # u/v advanced by the generator step
    l_d = net.discriminatorial_loss(net.discriminator(real_pyr), net.discriminator([a.detach() for a in att]))
    l_d.backward()
    for v in P2.values():
        if v.is_floating_point():
            v.requires_grad_(False)
    Pd = {k: (v.clone().requires_grad_(True) if (k.startswith("discriminator.") and v.is_floating_point() and not k.endswith(("_u", "_v"))) else v.clone())
          for k, v in P2.items()}
    want_d = R.discriminator_loss(Pd, [t.cpu() for t in real_pyr], att_c, noise[1], flip[1], noise[2], flip[2])
    want_d.backward()
    assert abs(l_d.item() - want_d.item()) <= 1e-4 * abs(want_d.item()) + 1e-6, (l_d.item(), want_d.item())
    dparams = dict(net.discriminator.named_parameters())
    for k, pm in dparams.items():
        w = Pd["discriminator." + k].grad
        err = (pm.grad.cpu() - w).abs().max().item()
        assert err <= 2e-3 * w.abs().max().item() + 1e-7, (k, err, w.abs().max().item())
    assert abs(l_d.item() - float(G["l_d_f64"])) <= 4 * abs(float(G["l_d"]) - float(G["l_d_f64"])) + 2e-4 * abs(float(G["l_d_f64"])), (l_d.item(), float(G["l_d"]))
    assert all(v.grad is None for v in net.segmentor.parameters()), "discriminator step must not touch the segmentor (att detached)"


def test_train_step_object_fp32_matches_manual_adam():
    """TrainStep (flat arenas, gradient sink, fused Adam) == the same losses + torch.optim.Adam on a twin."""
    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    from octave_amd import functional as F_
    dev = torch.device("cuda:0")
    Bn, H = 6, 48
    x, ys, real = _inputs(Bn, H, dev)

    def make():
        net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False, instance_noise=False, label_noise=False)
        fill_state_dict(net.state_dict())
        return net.to(dev).train()
    a, b = make(), make()
    try:
        step = TrainStep(a, lr=1e-3, compute_dtype=torch.float32)
        out = step(x, ys, mask_pyramid(real))
    finally:
        F_.set_grad_sink(False)
        from octave_amd.layers import defer_bn_counters
        defer_bn_counters(False)
    assert int(a.segmentor.encoder_0_1_2[1].num_batches_tracked) == 1
    # twin: plain autograd accumulation + torch Adam
    seg_params = [p for n, p in b.segmentor.named_parameters() if not n.startswith("linear_head_")]
    opt_s = torch.optim.Adam(seg_params, lr=1e-3)
    opt_d = torch.optim.Adam(b.discriminator.parameters(), lr=1e-3)
    att, agg, _ = b.segmentor(x)
    l = F_.wpce_dice(agg, ys, from_logits=True)
    p = F_.class_softmax(agg)
    loss = l[0] + l[1] + 0.1 * F_.interlayer_kl([p, *att], [1] * 5)[0] + 0.1 * F_.lsgan_generator(b.discriminator(att))
    opt_s.zero_grad(); opt_d.zero_grad()
    loss.backward()
    opt_s.step()
    opt_d.zero_grad()
    l_d = F_.lsgan_discriminator(b.discriminator(mask_pyramid(real)), b.discriminator([t.detach() for t in att]))
    l_d.backward()
    opt_d.step()
    # same kernels, same inputs: only the order of the float atomics differs between the two runs
    assert abs(out["loss_seg"].item() - loss.item()) < 2e-4 * abs(loss.item()) + 1e-6
    assert abs(out["loss_disc"].item() - l_d.item()) < 2e-4 * abs(l_d.item()) + 1e-6
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    worst = 0.0
    for k in pa:
        if "linear_head_" in k:
            assert torch.equal(pa[k], pb[k])
            continue
        d = (pa[k] - pb[k]).abs().max().item()
        worst = max(worst, d)
        # Adam's first step moves every weight by ~lr * sign(g): compare the updates, not the weights
        assert d <= 2.1e-3, (k, d)
    upd_a = torch.cat([(pa[k] - dict(make().named_parameters())[k]).flatten() for k in ("segmentor.fc.weight", "segmentor.decoder_0.conv.0.weight")])
    upd_b = torch.cat([(pb[k] - dict(make().named_parameters())[k]).flatten() for k in ("segmentor.fc.weight", "segmentor.decoder_0.conv.0.weight")])
    agree = (torch.sign(upd_a) == torch.sign(upd_b)).float().mean().item()
    assert agree > 0.99, agree


def test_train_step_bf16_runs_and_learns():
    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    from octave_amd import functional as F_
    dev = torch.device("cuda:0")
    Bn, H = 4, 64
    x, ys, real = _inputs(Bn, H, dev)
    torch.manual_seed(0)
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False).to(dev).train()
    try:
        step = TrainStep(net, lr=2e-4, compute_dtype=torch.bfloat16)
        losses = []
        for _ in range(6):
            out = step(x, ys, mask_pyramid(real))
            losses.append(out["wpce"].item() + out["dice"].item())
            assert all(np.isfinite(v.item()) for v in out.values())
    finally:
        F_.set_grad_sink(False)
        from octave_amd.layers import defer_bn_counters
        defer_bn_counters(False)
    print("[bf16 train] supervised loss per step:", ["%.4f" % v for v in losses])
    assert losses[-1] < losses[0]


def test_train_step_hipgraph_replay_matches_eager():
    """TrainStep.capture(): four hipGraphs around the two all-reduces.  One replay must reproduce the SAME step launched
    eagerly from the SAME state (weights, BN/spectral buffers, Adam moments, step counters, CPU generator state): the
    state is snapshotted after capture, the replay runs, the state is restored and the step runs again kernel by kernel.
    Only float-atomic ordering differs, so losses agree to 1e-4 relative; Adam turns a sign flip of a near-zero
    gradient into a 2*lr difference, hence the update bound.  (Comparing trajectories from different runs instead is
    ill-conditioned: two EAGER runs already differ by 0.3 % in g_adv after two steps with these weights.)"""
    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    from octave_amd import functional as F_
    dev = torch.device("cuda:0")
    Bn, H, lr = 6, 48, 1e-4
    x, ys, real = _inputs(Bn, H, dev)
    pyr = mask_pyramid(real)
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    st = TrainStep(net, lr=lr, compute_dtype=torch.float32)
    try:
        torch.manual_seed(11)
        st.capture(x, ys, pyr, warmup=1)          # one real (eager) step on the static buffers, then capture
        torch.cuda.synchronize()
        arenas = (st.seg_arena, st.disc_arena)
        snap_sd = {k: v.clone() for k, v in net.state_dict().items()}
        snap_mv = [(a.m.clone(), a.v.clone(), a.step_count) for a in arenas]
        rng = torch.get_rng_state()

        og = {k: v.clone() for k, v in st(x, ys, pyr).items()}
        torch.cuda.synchronize()
        pg = {k: v.detach().clone() for k, v in net.named_parameters()}
        bg = {k: v.clone() for k, v in net.named_buffers()}
        assert st.seg_arena.step_count == 2 and st.disc_arena.step_count == 2

        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(snap_sd[k])                # in place: parameters stay views of the arenas
        for a, (m, v, n) in zip(arenas, snap_mv):
            a.m.copy_(m), a.v.copy_(v)
            a.step_count = n
        torch.set_rng_state(rng)
        F_.bump_weight_epoch()
        oe = {k: v.clone() for k, v in st._eager_static().items()}
        torch.cuda.synchronize()
    finally:
        st.close()
    for k in oe:
        a, b = oe[k].item(), og[k].item()
        assert abs(a - b) <= 1e-4 * abs(a) + 1e-6, (k, a, b)
    pe = dict(net.named_parameters())
    worst, moved = 0.0, 0
    for k in pg:
        d = (pg[k] - pe[k]).abs()
        worst = max(worst, d.max().item())
        moved += int((d > 1e-6 + 1e-5 * pe[k].abs()).sum())
    total = sum(p.numel() for p in pg.values())
    print(f"[graph vs eager] max |dp| {worst:.2e} (2*lr = {2 * lr:.0e}); {moved}/{total} parameters differ")
    # measured 0.1-1.9 % in most runs, once 14.8 %: a large layer whose gradient is pure rounding noise takes +-lr steps either way
    assert worst <= 2.2 * lr and moved <= 0.30 * total, (worst, moved, total)
    for k in ("segmentor.encoder_0_1_2.1.running_mean", "segmentor.decoder_0.conv.1.running_var",
              "discriminator.spectral_dict.spectral_3.0.weight_u"):
        assert torch.allclose(bg[k].float(), dict(net.named_buffers())[k].float(), rtol=1e-4, atol=1e-6), k
    assert int(bg["segmentor.encoder_0_1_2.1.num_batches_tracked"]) == 2


def test_train_step_hipgraph_back_to_back_replays_stay_finite():
    """Replays enqueued WITHOUT host synchronisation (how bench.py and a real training loop drive the step): the host
    runs several steps ahead of the GPU, so the pinned staging rings, the device-side Adam corrections and every node
    of the graphs must stay ordered.  Regression test: hipMemsetAsync calls captured as memset NODES lost their order
    against the neighbouring kernel nodes in exactly this mode (accumulators cleared after the atomics that fill them;
    spectral-norm v = 0 -> 0/0 -> non-finite discriminator within a few steps).  The library now zero-fills with
    kernels (csrc/common.hpp: octa_zero_async); tools/nan_hunt.py is the step-by-step diagnostic."""
    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    dev = torch.device("cuda:0")
    Bn, H = 8, 128
    x, ys, real = _inputs(Bn, H, dev)
    pyr = mask_pyramid(real)
    torch.manual_seed(0)
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False).to(dev).train()
    st = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
    try:
        st.capture(x, ys, pyr)
        hist = []
        for _ in range(24):
            hist.append({k: v.clone() for k, v in st(x, ys, pyr).items()})
        torch.cuda.synchronize()
        for i, h in enumerate(hist):
            vals = {k: v.item() for k, v in h.items()}
            assert all(np.isfinite(v) for v in vals.values()), (i, vals)
            assert 0.0 <= vals["loss_disc"] < 50 and abs(vals["g_adv"]) < 50, (i, vals)
        assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
        assert st.seg_arena.step_count == 2 + 24
    finally:
        st.close()


def test_train_step_launch_paths_are_interchangeable():
    """After capture() the step can be replayed as hipGraphs or launched kernel by kernel on the same static buffers
    (TrainStep.launch / autotune_launch): both advance the same state (Adam step counters, BN counters, staged random draws)
    and can be mixed freely."""
    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    dev = torch.device("cuda:0")
    Bn, H = 4, 64
    x, ys, real = _inputs(Bn, H, dev)
    pyr = mask_pyramid(real)
    torch.manual_seed(0)
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False).to(dev).train()
    st = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
    try:
        st.capture(x, ys, pyr)
        n0 = st.seg_arena.step_count
        assert st.autotune_launch(x, ys, pyr, rounds=1, steps=1) in ("graph", "eager")
        assert st.seg_arena.step_count == n0 + 2
        for mode in ("eager", "graph", "eager", "graph"):
            st.launch = mode
            rng_before = torch.get_rng_state().clone()
            out = {k: v.clone() for k, v in st(x, ys, pyr).items()}
            torch.cuda.synchronize()
            assert all(np.isfinite(v.item()) for v in out.values()), (mode, out)
            # every step draws fresh discriminator noise from the global CPU generator, whichever path launches it
            assert not torch.equal(rng_before, torch.get_rng_state()), mode
        assert st.seg_arena.step_count == n0 + 6 and st.disc_arena.step_count == n0 + 6
        assert int(dict(net.named_buffers())["segmentor.encoder_0_1_2.1.num_batches_tracked"]) == n0 + 6
    finally:
        st.close()
