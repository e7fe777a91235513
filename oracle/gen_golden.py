"""Generate tests/golden/*.npz by running the REFERENCE on CPU (test infrastructure).

Run in the build container only (the reference never travels):

    python oracle/gen_golden.py            # needs /root/reference

The reference's ``segmentor/losses.py`` imports two packages that are not installed
here and cannot be (no network): ``kornia`` (one call: nearest ``resize``,
losses.py:126) and ``loguru`` (error logging only).  They are replaced in
``sys.modules`` by in-memory stand-ins before the import:
``resize -> F.interpolate(x, size, mode)`` and ``logger -> logging``.  The nearest
resize is therefore the one call whose parity is NOT pinned by the reference itself.

Weights and inputs are closed-form functions of (state_dict key, index) from
``oracle/fill.py``, so a fixture only stores the reference's OUTPUTS and gradients.
"""
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("OCTA_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.fill import fill_state_dict, hash_input  # noqa: E402


def _install_standins():
    k = types.ModuleType("kornia")
    kg = types.ModuleType("kornia.geometry")
    kt = types.ModuleType("kornia.geometry.transform")

    def resize(x, size, interpolation="nearest"):
        return F.interpolate(x, size=size, mode=interpolation)

    kt.resize = resize
    kg.transform = kt
    k.geometry = kg
    sys.modules.update({"kornia": k, "kornia.geometry": kg, "kornia.geometry.transform": kt})
    lg = types.ModuleType("loguru")
    lg.logger = logging.getLogger("reference")
    sys.modules["loguru"] = lg


def _np(t):
    return t.detach().cpu().numpy().copy()   # copy: buffers are updated in place by later calls


def _grads(mod):
    return {k: _np(p.grad) for k, p in mod.named_parameters() if p.grad is not None}


def _buffers(mod):
    return {k: _np(b) for k, b in mod.named_buffers()}


def _save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **d)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays)")


def block_fixture(tag, mod, x, d, extra_inputs=None):
    """Run mod(x) in train mode, backward against a fixed cotangent, record everything."""
    fill_state_dict(mod.state_dict(), salt=0)
    mod.train()
    x = x.clone().requires_grad_(True)
    out = mod(x)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    loss = 0
    for i, o in enumerate(outs):
        g = hash_input(tuple(o.shape), seed=7000 + i, lo=-1.0, hi=1.0)
        loss = loss + (o * g).sum()
        d[f"{tag}/out{i}"] = _np(o)
    loss.backward()
    d[f"{tag}/grad_x"] = _np(x.grad)
    for k, g in _grads(mod).items():
        d[f"{tag}/grad/{k}"] = g
    for k, b in _buffers(mod).items():
        d[f"{tag}/buf/{k}"] = b


def gen_blocks():
    from architectures.extra.resnest import Bottleneck, ResNestDecoder, SplAtConv2d, Upsampling
    from architectures.segmentor.blocks import AdversarialAttentionGate
    from torch import nn
    d = {}
    # encoder-style SplAt: cardinality 1, no conv bias (resnest.py:199-206)
    m = SplAtConv2d(16, 16, 3, padding=1, groups=1, bias=False, radix=2, norm_layer=nn.BatchNorm2d)
    block_fixture("splat_enc", m, hash_input((3, 16, 6, 5), 11, -1, 1), d)
    # decoder-style SplAt: cardinality 2, conv bias (resnest.py:27)
    m = SplAtConv2d(32, 32, 3, padding=1, stride=1, groups=2, radix=2, norm_layer=nn.BatchNorm2d)
    block_fixture("splat_dec", m, hash_input((3, 32, 5, 7), 12, -1, 1), d)
    # strided bottleneck with avg-down shortcut (resnest.py:381-394, avd 188-190)
    down = nn.Sequential(nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False),
                         nn.Conv2d(32, 64, 1, bias=False), nn.BatchNorm2d(64))
    m = Bottleneck(32, 16, stride=2, downsample=down, radix=2, cardinality=1, bottleneck_width=64,
                   avd=True, avd_first=False, norm_layer=nn.BatchNorm2d)
    block_fixture("bottleneck_s2", m, hash_input((3, 32, 8, 8), 13, -1, 1), d)
    # identity-shortcut bottleneck
    m = Bottleneck(64, 16, radix=2, cardinality=1, bottleneck_width=64, avd=True, avd_first=False,
                   norm_layer=nn.BatchNorm2d)
    block_fixture("bottleneck_id", m, hash_input((3, 64, 5, 5), 14, -1, 1), d)
    m = ResNestDecoder(64, 32)
    block_fixture("decoder", m, hash_input((3, 64, 6, 6), 15, -1, 1), d)
    m = Upsampling(16, 8)
    block_fixture("upsampling", m, hash_input((2, 16, 5, 3), 16, -1, 1), d)
    m = AdversarialAttentionGate(32, 2)
    block_fixture("aag", m, hash_input((2, 32, 7, 5), 17, -1, 1), d)
    m = AdversarialAttentionGate(16, 3)
    block_fixture("aag3", m, hash_input((2, 16, 4, 4), 18, -1, 1), d)
    _save("blocks.npz", d)


def gen_losses():
    from architectures.discriminator.losses import LSDiscriminatorialLoss, LSGeneratorLoss
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence, WeightedPartialCE
    d = {}
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

    wpce = WeightedPartialCE(num_classes=2, manual=True)
    cases = {
        "wpce_mean": dict(),
        "wpce_sum": dict(reduction="sum"),
        "wpce_full": dict(full=True),
        "wpce_ignore_bg": dict(ignore_bg=True),
    }
    for tag, kw in cases.items():
        p = probs(21).requires_grad_(True)
        ys = scribble(22)
        l = wpce(p, ys, **kw)
        l.backward()
        d[f"{tag}/loss"] = _np(l)
        d[f"{tag}/grad"] = _np(p.grad)
        d[f"{tag}/ys_after"] = _np(ys)
    p = probs(21).requires_grad_(True)
    ys = scribble(22, empty_class=1)       # ni = 0 for class 1 (eps path, losses.py:38)
    l = wpce(p, ys)
    l.backward()
    d["wpce_empty/loss"] = _np(l)
    d["wpce_empty/grad"] = _np(p.grad)

    p = probs(23).requires_grad_(True)
    t = scribble(24)
    l = DiceLoss()(p, t)
    l.backward()
    d["dice/loss"] = _np(l)
    d["dice/grad"] = _np(p.grad)
    # dense target (fully-supervised use, octa.py:54)
    p = probs(25).requires_grad_(True)
    t = F.one_hot((hash_input((B, H, W), 26) > 0.7).long(), 2).permute(0, 3, 1, 2).float()
    l = DiceLoss()(p, t)
    l.backward()
    d["dice_dense/loss"] = _np(l)
    d["dice_dense/grad"] = _np(p.grad)

    def pyramid(seed0, H0=32, n=6):
        return [probs(seed0 + i, (B, C, H0 >> max(i - 1, 0), H0 >> max(i - 1, 0))).requires_grad_(True)
                for i in range(n)]

    kl_cases = {
        "kl_default": (dict(), None),
        "kl_stopgrad": (dict(stop_gradient=True), None),
        "kl_weights": (dict(), [1, 0.5, 0, 2, 1]),
        "kl_short_weights": (dict(), [1, 2, 3, 4, 5, 6, 7]),
        "jsd": (dict(divergence="JSD"), None),
    }
    for tag, (kw, w) in kl_cases.items():
        att = pyramid(30)
        l = InterlayerDivergence(**kw)(att, w)
        l.backward()
        d[f"{tag}/loss"] = _np(l)
        for i, a in enumerate(att):
            d[f"{tag}/grad{i}"] = _np(a.grad) if a.grad is not None else np.zeros(tuple(a.shape), np.float32)

    r = hash_input((4, 1), 41, -2, 2).requires_grad_(True)
    f = hash_input((4, 1), 42, -2, 2).requires_grad_(True)
    l = LSDiscriminatorialLoss()(r, f)
    l.backward()
    d["lsd/loss"], d["lsd/grad_real"], d["lsd/grad_fake"] = _np(l), _np(r.grad), _np(f.grad)
    f2 = hash_input((4, 1), 43, -2, 2).requires_grad_(True)
    l = LSGeneratorLoss()(f2)
    l.backward()
    d["lsg/loss"], d["lsg/grad"] = _np(l), _np(f2.grad)
    _save("losses.npz", d)


def gen_discriminator():
    from architectures.discriminator.blocks import DiscriminatorBlock
    d = {}
    B, H = 2, 64
    shape = torch.Size((B, 2, H, H))
    m = DiscriminatorBlock(shape, is_training=True, depth=4, num_filters=8)
    fill_state_dict(m.state_dict())
    m.train()

    def pyr(seed):
        return [F.softmax(2 * hash_input((B, 2, H >> i, H >> i), seed + i, -1, 1), dim=1).requires_grad_(True)
                for i in range(5)]
    for call in range(2):
        torch.manual_seed(100 + call)
        noise = torch.normal(mean=0.0, std=0.2, size=(H, H))
        uni = torch.FloatTensor(1).uniform_(0, 1)
        torch.manual_seed(100 + call)            # replay: the forward consumes the same draws
        ys = pyr(50 + 10 * call)
        out = m(ys)
        g = hash_input(tuple(out.shape), 60 + call, -1, 1)
        m.zero_grad()
        (out * g).sum().backward()
        d[f"call{call}/noise"] = _np(noise)
        d[f"call{call}/uniform"] = _np(uni)
        d[f"call{call}/out"] = _np(out)
        for i, y in enumerate(ys):
            d[f"call{call}/grad_y{i}"] = _np(y.grad)
        for k, gr in _grads(m).items():
            d[f"call{call}/grad/{k}"] = gr
        for k, b in _buffers(m).items():
            d[f"call{call}/buf/{k}"] = b
    # eval mode: no power iteration, but noise / label noise still drawn (blocks.py:149-170)
    m.eval()
    torch.manual_seed(7)
    with torch.no_grad():
        out = m([y.detach() for y in pyr(90)])
    d["eval/out"] = _np(out)
    torch.manual_seed(7)
    d["eval/noise"] = _np(torch.normal(mean=0.0, std=0.2, size=(H, H)))
    d["eval/uniform"] = _np(torch.FloatTensor(1).uniform_(0, 1))
    _save("disc.npz", d)


def gen_unet():
    from architectures.models.octa import OctaScribbleNet
    for H in (48, 64):
        d = {}
        B = 3
        x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
        if H == 48:   # ResnestUNet.predict (compose.py:189-199) on its own instance: it runs a forward itself
            net0 = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
            fill_state_dict(net0.state_dict())
            net0.train()
            d["onehot"] = _np(net0.segmentor.predict(x, "one-hot")[1]).astype(np.uint8)
            d["sigmoid"] = _np(net0.segmentor.predict(x, "sigmoid")[1])
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
        fill_state_dict(net.state_dict())
        net.train()
        att, agg, x4 = net.segmentor(x)
        d["agg"] = _np(agg)
        d["x4"] = _np(x4)
        # the same reference modules in float64: how far the reference's OWN fp32 result is from the
        # exact answer (the network amplifies rounding noise through 93 train-mode BatchNorms; an
        # independent fp32 implementation cannot be closer to the fp32 reference than that)
        net64 = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
        fill_state_dict(net64.state_dict())
        net64 = net64.double().train()
        with torch.no_grad():
            att64, agg64, _ = net64.segmentor(x.double())
        d["agg_f64"] = _np(agg64)
        for i, a in enumerate(att64):
            d[f"att{i}_f64"] = _np(a)
        net64.zero_grad()
        att64, agg64, _ = net64.segmentor(x.double())     # with grad this time (running stats are not compared for net64)
        from architectures.segmentor.losses import DiceLoss as _Dice
        u64 = hash_input((B, 1, H, H), 4321)
        ys64 = torch.zeros(B, 2, H, H, dtype=torch.float64)
        ys64[:, 1:2] = (u64 < 0.05).double()
        ys64[:, 0:1] = ((u64 > 0.5) & (u64 < 0.55)).double()
        p64 = F.softmax(agg64, dim=1)
        loss64 = net64.supervised_loss(p64, ys64) + _Dice()(p64, ys64)
        loss64.backward()
        d["loss_f64"] = _np(loss64)
        for k, pr in net64.segmentor.named_parameters():
            if pr.grad is not None:
                d[f"gradnorm_f64/{k}"] = _np(pr.grad.norm())
        for i, a in enumerate(att):
            d[f"att{i}"] = _np(a)
        # segmentor-only loss (BASELINE config 2): WPCE + Dice on softmax(agg)
        from architectures.segmentor.losses import DiceLoss
        u = hash_input((B, 1, H, H), 4321)
        ys = torch.zeros(B, 2, H, H)
        ys[:, 1:2] = (u < 0.05).float()
        ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
        p = F.softmax(agg, dim=1)
        loss = net.supervised_loss(p, ys) + DiceLoss()(p, ys)
        loss.backward()
        d["loss"] = _np(loss)
        keep = ("encoder_0_1_2.0.0.weight", "encoder_0_1_2.1.weight", "encoder_1.0.conv2.conv.weight",
                "encoder_1.0.conv2.fc1.weight", "encoder_1.0.conv2.fc2.bias", "encoder_2.0.downsample.1.weight",
                "encoder_4.2.bn3.bias", "upsampling_4.up.bias", "upsampling_1.up.weight",
                "decoder_0.conv.0.weight", "decoder_0.conv.3.conv.weight", "decoder_0.conv.3.conv.bias",
                "decoder_1.downsample.0.weight", "aag_0.conv1.weight", "aag_3.conv1.bias", "fc.weight", "fc.bias")
        for k, pr in net.segmentor.named_parameters():
            if pr.grad is None:
                continue
            if k in keep:
                d[f"grad/{k}"] = _np(pr.grad)
            d[f"gradnorm/{k}"] = _np(pr.grad.double().norm())
        nograd = [k for k, pr in net.segmentor.named_parameters() if pr.grad is None]
        d["nograd_keys"] = np.array(nograd)
        for k in ("encoder_0_1_2.1.running_mean", "encoder_0_1_2.1.running_var",
                  "encoder_4.2.conv2.bn1.running_var", "decoder_0.conv.3.bn0.running_mean"):
            d[f"buf/{k}"] = _np(dict(net.segmentor.named_buffers())[k])
        _save(f"unet_{H}.npz", d)


def gen_trainstep():
    """One full adversarial step on the reference modules, B=6, 48x48: the two losses and the
    gradient norms, in fp32 and (same modules, .double()) in float64, so that the HIP path and the oracle
    can be pinned relative to the reference's own rounding band."""
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    B, H = 6, 48
    d = {}
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    u = hash_input((B, 1, H, H), 4321)
    ys = torch.zeros(B, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    dense = (hash_input((B, H, H), 999) > 0.8).long()
    real = F.one_hot(dense, 2).permute(0, 3, 1, 2).float()
    # RNG draws consumed, in order: 3 discriminator calls x (normal(H,W), uniform(1))
    torch.manual_seed(2024)
    for c in range(3):
        d[f"noise{c}"] = _np(torch.normal(mean=0.0, std=0.2, size=(H, H)))
        d[f"uniform{c}"] = _np(torch.FloatTensor(1).uniform_(0, 1))
    for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
        fill_state_dict(net.state_dict())
        net = net.to(dt).train()
        xx, yy = x.to(dt), ys.to(dt)
        real_pyr = [real[:, :, ::2 ** i, ::2 ** i].contiguous().to(dt) for i in range(5)]
        torch.manual_seed(2024)
        att, agg, _ = net.segmentor(xx)
        p = F.softmax(agg, dim=1)
        parts = [net.supervised_loss(p, yy), DiceLoss()(p, yy), InterlayerDivergence()([p, *att]), net.generator_loss(net.discriminator(att))]
        l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
        net.zero_grad()
        l_seg.backward()
        d["l_seg" + tag] = _np(l_seg)
        d["parts" + tag] = np.array([float(v) for v in parts])
        for k, pr in net.segmentor.named_parameters():
            if pr.grad is not None:
                d[f"seg_gradnorm{tag}/{k}"] = _np(pr.grad.double().norm())
                if k in grads:
                    d[f"seg_grad{tag}/{k}"] = _np(pr.grad.contiguous().flatten()[::grad_stride(pr.numel())])
        net.zero_grad()
        l_d = net.discriminatorial_loss(net.discriminator(real_pyr), net.discriminator([a.detach() for a in att]))
        l_d.backward()
        d["l_d" + tag] = _np(l_d)
        for k, pr in net.discriminator.named_parameters():
            d[f"disc_gradnorm{tag}/{k}"] = _np(pr.grad.double().norm())
            if pr.numel() <= 4096 and tag == "":
                d[f"disc_grad/{k}"] = _np(pr.grad)
    _save("trainstep_48.npz", d)


def gen_extras():
    """Fixtures for the rest of the public surface and for parity at a BASELINE size:
    reference state_dict layouts, eval-mode inference (logits / softmax / sigmoid / one-hot), classification heads,
    the dual-head U-Nets, the non-manual WPCE branches, LabelNoise 'label', InstanceNoise without clipping."""
    from architectures.discriminator.blocks import InstanceNoise, LabelNoise
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.compose import ResnestUnetParallelHead, ResnestUnetParallelHeadAttentionGate
    from architectures.segmentor.losses import WeightedPartialCE
    d = {}
    # (1) state_dict layouts dumped from the reference itself
    net = OctaScribbleNet(torch.Size((2, 3, 48, 48)), torch.Size((2, 2, 48, 48)), True, False)
    sd = net.state_dict()
    d["layout/octa/keys"] = np.array(list(sd.keys()))
    d["layout/octa/shapes"] = np.array([",".join(str(v) for v in t.shape) for t in sd.values()])
    for tag, cls in (("ph", ResnestUnetParallelHead), ("phag", ResnestUnetParallelHeadAttentionGate)):
        m = cls(2, False)
        sdm = m.state_dict()
        d[f"layout/{tag}/keys"] = np.array(list(sdm.keys()))
        d[f"layout/{tag}/shapes"] = np.array([",".join(str(v) for v in t.shape) for t in sdm.values()])
    # (2) eval-mode inference, 48 (odd H/16) and 64 (even)
    for H in (48, 64):
        B = 2
        x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), False, False)
        fill_state_dict(net.state_dict())
        net.eval()
        with torch.no_grad():
            att, agg, x4 = net.segmentor(x)
            d[f"eval{H}/agg"] = _np(agg)
            for i, a in enumerate(att):
                d[f"eval{H}/att{i}"] = _np(a)
            d[f"eval{H}/softmax"] = _np(net.segmentor.predict(x, "softmax")[1])
            d[f"eval{H}/sigmoid"] = _np(net.segmentor.predict(x, "sigmoid")[1])
            d[f"eval{H}/onehot"] = _np(net.segmentor.predict(x, "one-hot")[1]).astype(np.uint8)
            net64 = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), False, False)
            fill_state_dict(net64.state_dict())
            net64 = net64.double().eval()
            d[f"eval{H}/agg_f64"] = _np(net64.segmentor(x.double())[1])
            if H == 48:
                # (3) classification heads (compose.py:201-230)
                for mode in ("classic", "ae-squash", "ae-extract"):
                    for method in ("softmax", "sigmoid"):
                        cp, _, pred = net.segmentor.classification_predict(x, method, mode)
                        d[f"cls/{mode}/{method}"] = _np(cp)
                d["cls/predicate"] = _np(pred)
    # (4) dual-head U-Nets, train mode, B = 3, 48x48, fp32 + float64 twin
    B, H = 3, 48
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    for tag, cls, kw in (("ph", ResnestUnetParallelHead, {}), ("phag", ResnestUnetParallelHeadAttentionGate, {"gating_leveL": 3})):
        for suffix, dt in (("", torch.float32), ("_f64", torch.float64)):
            m = cls(2, False, **kw)
            fill_state_dict(m.state_dict())
            m = m.to(dt).train()
            with torch.no_grad():
                out = m(x.to(dt))
            if tag == "ph":
                d[f"{tag}/agg{suffix}"] = _np(out)
            else:
                (att, att_c), agg = out
                d[f"{tag}/agg{suffix}"] = _np(agg)
                d[f"{tag}/n_att"] = np.array([len(att), len(att_c)])
                for i, a in enumerate(att):
                    d[f"{tag}/att{i}{suffix}"] = _np(a)
                for i, a in enumerate(att_c):
                    d[f"{tag}/att_c{i}{suffix}"] = _np(a)
    # (5) WPCE non-manual branches
    Bq, C, Hq, Wq = 3, 2, 16, 16
    p = F.softmax(3.0 * hash_input((Bq, C, Hq, Wq), 21, -1, 1), dim=1).requires_grad_(True)
    u = hash_input((Bq, 1, Hq, Wq), 22)
    ys = torch.zeros(Bq, C, Hq, Wq)
    ys[:, 1:2] = (u < 0.08).float()
    ys[:, 0:1] = ((u > 0.3) & (u < 0.4)).float()
    for tag, kw in (("ce", {}), ("ce_full", {"full": True})):
        pp = p.detach().clone().requires_grad_(True)
        l = WeightedPartialCE(num_classes=2, manual=False)(pp, ys.clone(), **kw)
        l.backward()
        d[f"wpce_{tag}/loss"], d[f"wpce_{tag}/grad"] = _np(l), _np(pp.grad)
    z1 = (2.0 * hash_input((Bq, 1, Hq, Wq), 27, -1, 1)).requires_grad_(True)
    t1 = (hash_input((Bq, 1, Hq, Wq), 28) < 0.3).float()
    l = WeightedPartialCE(num_classes=1, manual=True)(z1, t1)
    l.backward()
    d["wpce_bce/loss"], d["wpce_bce/grad"] = _np(l), _np(z1.grad)
    # (6) LabelNoise 'label' (flip forced / never) and InstanceNoise without clipping
    xl = hash_input((4, 1), 44, -0.5, 1.5).requires_grad_(True)
    yl = LabelNoise(prob=2.0, mode="label")(xl)
    (yl * hash_input((4, 1), 45, -1, 1)).sum().backward()
    d["labelflip/out"], d["labelflip/grad"] = _np(yl), _np(xl.grad)
    d["labelkeep/out"] = _np(LabelNoise(prob=-1.0, mode="label")(xl))
    torch.manual_seed(11)
    xin = hash_input((2, 2, 8, 8), 46, -0.2, 1.2)
    d["noise_noclip/out"] = _np(InstanceNoise(torch.Size((2, 2, 8, 8)), 0.0, 0.2, False, True)(xin))
    torch.manual_seed(11)
    d["noise_noclip/noise"] = _np(torch.normal(mean=0.0, std=0.2, size=(8, 8)))
    _save("extras.npz", d)


def gen_unet304():
    """BASELINE config-1 size: B = 2, 304 x 304, fp32, train mode -- logits and their float64 twin (stored as float32: its
    rounding, 2e-6 absolute, is three orders below the reference's own fp32-vs-fp64 band)."""
    from architectures.models.octa import OctaScribbleNet
    B, H = 2, 304
    d = {}
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
    fill_state_dict(net.state_dict())
    net.train()
    with torch.no_grad():
        d["agg"] = _np(net.segmentor(x)[1])
    net64 = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
    fill_state_dict(net64.state_dict())
    net64 = net64.double().train()
    with torch.no_grad():
        d["agg_f64_as_f32"] = _np(net64.segmentor(x.double())[1]).astype(np.float32)
    _save("unet_304.npz", d)



# Full gradients stored by the conditioned 400 x 400 fixture: sixteen tensors spread over all ten gradient buckets of the train step
# (octave_amd/train.py SEG_GRAD_ORDER), every kernel family of the weight gradients among them (stem 3 -> 32, grouped split-attention
# convs, the big decoder 3x3s, the up-shuffle, the attention micro-net, BatchNorm affine, the 1x1 gates / head).  Tensors beyond
# 40 000 elements are stored as flat[::stride] of the OIHW-logical order.
GRAD_TENSORS_400C = (
    "fc.weight", "decoder_0.conv.0.weight", "decoder_0.conv.3.conv.weight", "upsampling_0.up.weight", "decoder_1.conv.3.conv.weight",
    "aag_2.conv1.weight", "decoder_2.conv.3.fc1.weight", "decoder_2.conv.0.weight", "decoder_3.conv.3.bn0.weight", "decoder_4.conv.0.weight",
    "encoder_4.2.conv3.weight", "encoder_3.0.conv2.fc1.weight", "encoder_2.0.conv1.weight", "encoder_1.0.conv2.conv.weight",
    "encoder_0_1_2.0.3.weight", "encoder_0_1_2.0.0.weight")


from oracle.gen_golden_meta import grad_stride  # noqa: E402


def _adversarial_step_fixture(B, H, d, noise_seed=2024, logits=False, scale=None, grads=(), logits_sub=1):
    """One full adversarial step on the reference modules at (B, H): the four loss parts, both losses and every
    gradient norm, in fp32 and (same modules, .double()) in float64.  logits=True also stores the segmentor's logits (the
    float64 twin rounded to float32: 2e-6 absolute, three orders below the reference's own fp32-vs-fp64 band)."""
    from architectures.models.octa import OctaScribbleNet
    from architectures.segmentor.losses import DiceLoss, InterlayerDivergence
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    u = hash_input((B, 1, H, H), 4321)
    ys = torch.zeros(B, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    dense = (hash_input((B, H, H), 999) > 0.8).long()
    real = F.one_hot(dense, 2).permute(0, 3, 1, 2).float()
    torch.manual_seed(noise_seed)
    for c in range(3):
        d[f"noise{c}"] = _np(torch.normal(mean=0.0, std=0.2, size=(H, H)))
        d[f"uniform{c}"] = _np(torch.FloatTensor(1).uniform_(0, 1))
    for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
        fill_state_dict(net.state_dict(), scale=scale)
        net = net.to(dt).train()
        xx, yy = x.to(dt), ys.to(dt)
        real_pyr = [real[:, :, ::2 ** i, ::2 ** i].contiguous().to(dt) for i in range(5)]
        torch.manual_seed(noise_seed)
        att, agg, _ = net.segmentor(xx)
        if logits:
            d["agg" + ("_f64_as_f32" if tag else "")] = _np(agg[:, :, ::logits_sub, ::logits_sub]).astype(np.float32)
        p = F.softmax(agg, dim=1)
        parts = [net.supervised_loss(p, yy), DiceLoss()(p, yy), InterlayerDivergence()([p, *att]), net.generator_loss(net.discriminator(att))]
        l_seg = parts[0] + parts[1] + 0.1 * parts[2] + 0.1 * parts[3]
        net.zero_grad()
        l_seg.backward()
        d["l_seg" + tag] = _np(l_seg)
        d["parts" + tag] = np.array([float(v) for v in parts])
        for k, pr in net.segmentor.named_parameters():
            if pr.grad is not None:
                d[f"seg_gradnorm{tag}/{k}"] = _np(pr.grad.double().norm())
                if k in grads:
                    d[f"seg_grad{tag}/{k}"] = _np(pr.grad.contiguous().flatten()[::grad_stride(pr.numel())])
        net.zero_grad()
        l_d = net.discriminatorial_loss(net.discriminator(real_pyr), net.discriminator([a.detach() for a in att]))
        l_d.backward()
        d["l_d" + tag] = _np(l_d)
        for k, pr in net.discriminator.named_parameters():
            d[f"disc_gradnorm{tag}/{k}"] = _np(pr.grad.double().norm())


def gen_trainstep400():
    """The HEADLINE resolution (BASELINE configs[2]: 400 x 400, H/16 = 25 odd -> pad / crop compose.py:122-130,142-147, 12 x 12
    discriminator head blocks.py:68-72) end to end on the reference modules at B = 2 (fp32 + float64 twin): logits, loss parts,
    both losses, every gradient norm and the consumed CPU random draws."""
    d = {}
    _adversarial_step_fixture(2, 400, d, logits=True)
    _save("trainstep_400.npz", d)


def gen_trainstep400c():
    """The WELL-CONDITIONED twin of trainstep_400: B = 4 (every split-attention bn1 normalises over four samples instead of two,
    extra/resnest.py:120-121) and the discriminator's full-extent head scaled by COND_SCALE so that the LS-GAN generator term is O(1)
    (g_adv ~ 1 instead of 180 of a loss of ~20; discriminator/blocks.py:68-72).  Everything else as trainstep_400; the logits are
    stored on every second row / column, and sixteen FULL gradients (GRAD_TENSORS_400C) beside every gradient norm."""
    from oracle.fill import COND_SCALE
    d = {}
    _adversarial_step_fixture(4, 400, d, logits=True, scale=COND_SCALE, grads=GRAD_TENSORS_400C, logits_sub=2)
    d["cond_scale_head"] = np.array([COND_SCALE["discriminator.out.0.weight"]])
    _save("trainstep_400c.npz", d)


def gen_round4():
    """Eval-mode inference of the reference at the two BASELINE resolutions (B = 2, 304 x 304 and 400 x 400, closed-form
    weights): ``ResnestUNet.predict(x, 'one-hot')`` (compose.py:189-199) as a packed bit mask of class 1, plus the packed mask
    of pixels whose float64 logit margin exceeds 1e-4 x the logit scale (where the argmax is decidable: the HIP path must be
    bit-exact there).  Used by the ``dice_vs_ref`` figure of bench.py and by tests/test_round4.py."""
    from architectures.models.octa import OctaScribbleNet
    d = {}
    for H in (304, 400):
        B = 2
        x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), False, False)
        fill_state_dict(net.state_dict())
        net.eval()
        net64 = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), False, False)
        fill_state_dict(net64.state_dict())
        net64 = net64.double().eval()
        with torch.no_grad():
            onehot = net.segmentor.predict(x, "one-hot")[1]
            agg = net.segmentor(x)[1]
            agg64 = net64.segmentor(x.double())[1]
        assert onehot.shape == (B, 2, H, H)
        cls1 = _np(onehot[:, 1]).astype(np.uint8)
        assert ((cls1 == 1) == (_np(onehot[:, 0]) == 0)).all()
        scale = float(agg64.abs().max())
        margin = (agg64[:, 1] - agg64[:, 0]).abs()
        d[f"eval{H}/onehot_cls1_bits"] = np.packbits(cls1.reshape(-1))
        d[f"eval{H}/decidable_bits"] = np.packbits((_np(margin) > 1e-4 * scale).astype(np.uint8).reshape(-1))
        d[f"eval{H}/shape"] = np.array([B, H, H])
        d[f"eval{H}/logit_scale"] = np.array([scale])
        d[f"eval{H}/ref32_vs_ref64_maxabs"] = np.array([float((agg.double() - agg64).abs().max())])
        d[f"eval{H}/agg_f64_sub8"] = _np(agg64[:, :, ::8, ::8]).astype(np.float32)
        d[f"eval{H}/frac_cls1"] = np.array([float(cls1.mean())])
    _save("round4.npz", d)


def gen_trainstep304():
    """BASELINE configs[0] end to end: the full adversarial step at B = 2, 304 x 304 on the reference modules (fp32 + float64
    twin): loss parts, both losses, every gradient norm and the consumed CPU random draws."""
    d = {}
    _adversarial_step_fixture(2, 304, d)
    _save("trainstep_304.npz", d)


def gen_round3():
    """(1) the stand-alone ResNet checkpoint layout of resnest50() (extra/resnest.py:451-459: what resnest50-528c19ca.pth
    holds and what load_state_dict(torch.load(model_path)) expects), dumped from the reference; (2) BACKWARD fixtures for the two
    dual-head U-Nets (segmentor/compose.py:233-362, 365-527) at B = 3, 48 x 48, train mode: cotangents are closed-form
    (hash_input, seeds 7100 + output index), every parameter's gradient norm is stored in fp32 and float64, a few gradients
    in full."""
    from architectures.extra.resnest import resnest50
    from architectures.segmentor.compose import ResnestUnetParallelHead, ResnestUnetParallelHeadAttentionGate
    d = {}
    sd = resnest50().state_dict()
    d["layout/resnest50/keys"] = np.array(list(sd.keys()))
    d["layout/resnest50/shapes"] = np.array([",".join(str(v) for v in t.shape) for t in sd.values()])
    B, H = 3, 48
    x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
    full = ("decoder_4_c.conv.0.weight", "decoder_0_c.conv.3.conv.bias", "upsampling_4_c.up.weight", "upsampling_0_c.up.bias", "fc_c.weight", "fc_c.bias",
            "fc.weight", "aag_3_c.conv1.weight", "aag_0_c.conv1.bias", "aag_3.conv1.weight", "encoder_4.2.bn3.bias", "encoder_1.0.conv1.weight",
            "encoder_0_1_2.0.0.weight", "decoder_1_c.conv.0.weight", "upsampling_1_c.up.weight", "aag_1_c.conv1.weight")
    for tag, cls, kw in (("ph", ResnestUnetParallelHead, {}), ("phag", ResnestUnetParallelHeadAttentionGate, {"gating_leveL": 3})):
        for suffix, dt in (("", torch.float32), ("_f64", torch.float64)):
            m = cls(2, False, **kw)
            fill_state_dict(m.state_dict())
            m = m.to(dt).train()
            out = m(x.to(dt))
            outs = [out] if tag == "ph" else [out[1], *out[0][0], *out[0][1]]
            loss = 0
            for i, o in enumerate(outs):
                loss = loss + (o * hash_input(tuple(o.shape), 7100 + i, -1.0, 1.0).to(dt)).sum()
            loss.backward()
            d[f"{tag}/loss{suffix}"] = _np(loss)
            d[f"{tag}/n_out"] = np.array([len(outs)])
            for k, pr in m.named_parameters():
                if pr.grad is not None:
                    d[f"{tag}/gradnorm{suffix}/{k}"] = _np(pr.grad.double().norm())
                    if suffix == "" and k in full:
                        d[f"{tag}/grad/{k}"] = _np(pr.grad)
            if suffix == "":
                d[f"{tag}/nograd_keys"] = np.array([k for k, pr in m.named_parameters() if pr.grad is None])
    _save("round3.npz", d)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(f"reference not found at {REF}")
    sys.path.insert(0, REF)
    _install_standins()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["blocks", "losses", "disc", "unet", "trainstep", "extras", "unet304", "trainstep304", "round3",
                             "trainstep400", "round4", "trainstep400c"]
    if "blocks" in which:
        gen_blocks()
    if "losses" in which:
        gen_losses()
    if "disc" in which:
        gen_discriminator()
    if "unet" in which:
        gen_unet()
    if "trainstep" in which:
        gen_trainstep()
    if "extras" in which:
        gen_extras()
    if "unet304" in which:
        gen_unet304()
    if "trainstep304" in which:
        gen_trainstep304()
    if "round3" in which:
        gen_round3()
    if "trainstep400" in which:
        gen_trainstep400()
    if "trainstep400c" in which:
        gen_trainstep400c()
    if "round4" in which:
        gen_round4()
