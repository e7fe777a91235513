"""CPU restatement (plain torch, functional) of the OCTAve hot path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Every function cites the
reference file:line it follows (paths relative to /root/reference/architectures).
State lives in a flat ``dict[str, Tensor]`` keyed exactly like the reference's
``state_dict()`` so fixtures and checkpoints map 1:1.  Gradients come from
torch autograd on CPU.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _k(prefix: str, name: str) -> str:
    return f"{prefix}.{name}" if prefix else name

# resnest50(): layers [3,4,6,3], radix 2, cardinality 1, deep stem 32, avg_down, avd
# (extra/resnest.py:451-455)
STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_PLANES = (64, 128, 256, 512)


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def batch_norm(x: Tensor, P: State, prefix: str, training: bool = True) -> Tensor:
    """nn.BatchNorm2d, momentum 0.1, eps 1e-5; updates running stats in place when
    training (e.g. extra/resnest.py:25,86,90,182,224,338,393)."""
    rm, rv = P[_k(prefix, "running_mean")], P[_k(prefix, "running_var")]
    out = F.batch_norm(x, rm, rv, P[_k(prefix, "weight")], P[_k(prefix, "bias")],
                       training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training and (_k(prefix, "num_batches_tracked")) in P:
        P[_k(prefix, "num_batches_tracked")] += 1
    return out


def splat_conv2d(x: Tensor, P: State, prefix: str, cardinality: int, radix: int = 2,
                 training: bool = True) -> Tensor:
    """SplAtConv2d.forward, extra/resnest.py:97-138 (ctor 62-95), 3x3 s1 p1."""
    bias = P.get(_k(prefix, "conv.bias"))
    x = F.conv2d(x, P[_k(prefix, "conv.weight")], bias, stride=1, padding=1,
                 groups=cardinality * radix)                              # :99
    x = F.relu(batch_norm(x, P, _k(prefix, "bn0"), training))               # :101,105
    B, RC = x.shape[:2]
    C = RC // radix
    splits = torch.split(x, C, dim=1)                                     # :109
    gap = sum(splits)                                                     # :111
    gap = F.adaptive_avg_pool2d(gap, 1)                                   # :116
    gap = F.conv2d(gap, P[_k(prefix, "fc1.weight")], P[_k(prefix, "fc1.bias")], groups=cardinality)  # :118
    gap = F.relu(batch_norm(gap, P, _k(prefix, "bn1"), training))           # :121-122
    att = F.conv2d(gap, P[_k(prefix, "fc2.weight")], P[_k(prefix, "fc2.bias")], groups=cardinality)  # :125
    att = att.view(B, radix, C)            # NOTE: no cardinality transpose (reference, not upstream)
    att = F.softmax(att, dim=1).view(B, -1, 1, 1)                         # :127
    atts = torch.split(att, C, dim=1)                                     # :133
    return sum(a * s for a, s in zip(atts, splits)).contiguous()          # :135,138


def bottleneck(x: Tensor, P: State, prefix: str, stride: int, has_down: bool,
               training: bool = True) -> Tensor:
    """Bottleneck.forward, extra/resnest.py:234-267 for the resnest50 configuration
    (radix 2, cardinality 1, avd, avd_first False, avg_down)."""
    out = F.conv2d(x, P[_k(prefix, "conv1.weight")])                        # :237
    out = F.relu(batch_norm(out, P, _k(prefix, "bn1"), training))           # :238,241
    out = splat_conv2d(out, P, _k(prefix, "conv2"), cardinality=1, training=training)  # :246
    if stride > 1:                                                        # avd = stride>1 or is_first(False) :185
        out = F.avg_pool2d(out, 3, stride, padding=1)                     # :189,253-254
    out = F.conv2d(out, P[_k(prefix, "conv3.weight")])                      # :256
    out = batch_norm(out, P, _k(prefix, "bn3"), training)                   # :257
    if has_down:                                                          # :381-394
        r = x
        if stride > 1:
            r = F.avg_pool2d(r, stride, stride, ceil_mode=True, count_include_pad=False)
        # stride 1: AvgPool2d(1,1) == identity (layer1 block 0)
        r = F.conv2d(r, P[_k(prefix, "downsample.1.weight")])
        r = batch_norm(r, P, _k(prefix, "downsample.2"), training)
    else:
        r = x
    return F.relu(out + r)                                                # :264-265


def encoder_stage(x: Tensor, P: State, prefix: str, idx: int, training: bool = True) -> Tensor:
    """_make_layer result (extra/resnest.py:376-429): first block strided + downsample."""
    stride = 1 if idx == 0 else 2
    for b in range(STAGE_BLOCKS[idx]):
        x = bottleneck(x, P, _k(prefix, str(b)), stride if b == 0 else 1, b == 0, training)
    return x


def stem(x: Tensor, P: State, prefix: str, training: bool = True) -> Tensor:
    """encoder_0_1_2 = Sequential(deep-stem conv1, bn1, relu): extra/resnest.py:325-339,
    segmentor/compose.py:40-44."""
    x = F.conv2d(x, P[_k(prefix, "0.0.weight")], stride=2, padding=1)
    x = F.relu(batch_norm(x, P, _k(prefix, "0.1"), training))
    x = F.conv2d(x, P[_k(prefix, "0.3.weight")], padding=1)
    x = F.relu(batch_norm(x, P, _k(prefix, "0.4"), training))
    x = F.conv2d(x, P[_k(prefix, "0.6.weight")], padding=1)
    return F.relu(batch_norm(x, P, _k(prefix, "1"), training))


def resnest_decoder(x: Tensor, P: State, prefix: str, training: bool = True) -> Tensor:
    """ResNestDecoder.forward, extra/resnest.py:38-43 (ctor 18-36)."""
    r = F.conv2d(x, P[_k(prefix, "downsample.0.weight")])
    r = batch_norm(r, P, _k(prefix, "downsample.1"), training)
    o = F.conv2d(x, P[_k(prefix, "conv.0.weight")], padding=1)
    o = F.relu(batch_norm(o, P, _k(prefix, "conv.1"), training))
    o = F.relu(splat_conv2d(o, P, _k(prefix, "conv.3"), cardinality=2, training=training))
    return F.relu(r + o)


def upsampling(x: Tensor, P: State, prefix: str) -> Tensor:
    """Upsampling.forward, extra/resnest.py:46-54: ConvTranspose2d k2 s2 + bias."""
    return F.conv_transpose2d(x, P[_k(prefix, "up.weight")], P[_k(prefix, "up.bias")], stride=2)


def attention_gate(x: Tensor, P: State, prefix: str) -> Tuple[Tensor, Tensor]:
    """AdversarialAttentionGate.forward, segmentor/blocks.py:38-46."""
    y = F.softmax(F.conv2d(x, P[_k(prefix, "conv1.weight")], P[_k(prefix, "conv1.bias")]), dim=1)
    mask = y[:, 1:].sum(dim=1, keepdim=True)
    return x * mask, y


# --------------------------------------------------------------------------- #
# segmentor
# --------------------------------------------------------------------------- #
def resnest_unet_forward(x: Tensor, P: State, prefix: str = "segmentor", gating_level: int = 4,
                         training: bool = True):
    """ResnestUNet.forward, segmentor/compose.py:100-187 (encoder_gating=False)."""
    p = prefix + "." if prefix else ""
    x_0_0 = stem(x, P, p + "encoder_0_1_2", training)                     # :102
    x_0_1 = F.max_pool2d(x_0_0, 3, 2, 1)                                  # :103
    x_1 = encoder_stage(x_0_1, P, p + "encoder_1", 0, training)           # :109
    x_2 = encoder_stage(x_1, P, p + "encoder_2", 1, training)             # :113
    x_3 = encoder_stage(x_2, P, p + "encoder_3", 2, training)             # :117
    pad_h = x_3.shape[2] % 2 == 1                                         # :125-130
    pad_w = x_3.shape[3] % 2 == 1
    if pad_h or pad_w:
        x_3 = F.pad(x_3, (0, int(pad_w), 0, int(pad_h)))
    x_4 = encoder_stage(x_3, P, p + "encoder_4", 3, training)             # :132

    att: List[Tensor] = []
    d = upsampling(x_4, P, p + "upsampling_4")                            # :140
    d = torch.cat((x_3, d), dim=1)                                        # :141
    d = d[:, :, :d.shape[2] - int(pad_h), :d.shape[3] - int(pad_w)]       # :142-147
    d = resnest_decoder(d, P, p + "decoder_4", training)                  # :149
    if gating_level >= 4:
        d, y = attention_gate(d, P, p + "aag_4"); att.append(y)           # :150-152
    for lvl, skip in ((3, x_2), (2, x_1), (1, x_0_0)):                    # :154-173
        d = upsampling(d, P, p + f"upsampling_{lvl}")
        d = torch.cat((skip, d), dim=1)
        d = resnest_decoder(d, P, p + f"decoder_{lvl}", training)
        if gating_level >= lvl:
            d, y = attention_gate(d, P, p + f"aag_{lvl}"); att.append(y)
    d = upsampling(d, P, p + "upsampling_0")                              # :175
    d = resnest_decoder(d, P, p + "decoder_0", training)                  # :176
    if gating_level >= 0:
        d, y = attention_gate(d, P, p + "aag_0"); att.append(y)           # :177-179
    agg = F.conv2d(d, P[p + "fc.weight"], P[p + "fc.bias"])               # :181
    att.reverse()                                                         # :183
    return tuple(att), agg, x_4                                           # :187


def predict(agg_map: Tensor, method: str = "softmax") -> Tensor:
    """ResnestUNet.predict post-processing, segmentor/compose.py:189-199."""
    if method == "softmax":
        return F.softmax(agg_map, dim=1)
    if method == "sigmoid":
        return torch.sigmoid(agg_map)
    if method == "one-hot":
        return F.one_hot(torch.argmax(agg_map, dim=1)).permute(0, 3, 1, 2)
    return agg_map


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def weighted_partial_ce(y_hat: Tensor, ys: Tensor, num_classes: int, ignore_bg: bool = False,
                        reduction: str = "mean", full: bool = False) -> Tensor:
    """WeightedPartialCE.forward with manual=True (as built by models/octa.py:52),
    segmentor/losses.py:26-61.  y_hat are probabilities.  `ignore_bg` zeroes
    ys[:,0] IN PLACE like the reference (:29-30)."""
    assert y_hat.shape[1] == ys.shape[1], "Number of class mismatch."
    if ignore_bg:
        ys[:, 0] = 0
    if not full:
        y_hat = y_hat * ys                                                # :32
    ni = ys.sum(dim=(0, 2, 3))                                            # :35
    n_tot = ni.sum()                                                      # :37
    w = n_tot / (ni + 1e-12)                                              # :38
    per_pix = -(w.view(1, -1, 1, 1) * ys * torch.log(y_hat + 1e-12)).sum(dim=1)  # :52-54
    return per_pix.mean() if reduction == "mean" else per_pix.sum()       # :55


def dice_loss(inp: Tensor, target: Tensor, eps: float = 1e-12) -> Tensor:
    """DiceLoss.forward, segmentor/losses.py:70-74."""
    inter = (inp * target).sum(dim=(1, 2, 3))
    card = (inp + target).sum(dim=(1, 2, 3))
    return (1.0 - 2.0 * inter / (card + eps)).mean()


def nearest_resize(a: Tensor, size: Tuple[int, int]) -> Tensor:
    """kornia.geometry.transform.resize(..., interpolation='nearest') stand-in
    (segmentor/losses.py:126; kornia absent -> parity unpinned for this call).
    For the integer ratios of the hot path: src = dst // f."""
    return F.interpolate(a, size=size, mode="nearest")


def interlayer_divergence(attentions: Sequence[Tensor], weights: Optional[list] = None,
                          stop_gradient: bool = False, divergence: str = "KLD",
                          mode: str = "mean", eps: float = 1e-12) -> Tensor:
    """InterlayerDivergence.forward, segmentor/losses.py:111-172."""
    basis = attentions[0].detach() if stop_gradient else attentions[0]    # :114
    H, W = basis.shape[2:]
    rest = list(attentions[1:])
    if weights is None:
        weights = [1] * len(rest)                                         # :118-119
    elif len(weights) != len(rest):
        weights = weights[:len(attentions)]                               # :121-123 (truncate)
    post = [nearest_resize(a, (H, W)) * w for a, w in zip(rest, weights) if w != 0]  # :124-126
    post = torch.stack(post, 0)
    if divergence == "KLD":
        if mode != "mean":
            raise NotImplementedError("Not implemented yet.")             # :149-150
        m_log = torch.log(post + 1e-12).sum(0) / sum(weights)             # :135
        div = (basis * (torch.log(basis + 1e-12) - m_log)).sum(dim=1).mean()  # :137-139
        if torch.isnan(div).any():
            raise Exception("Divergence is NaN")                          # :140-142
        return div
    if divergence == "JSD":                                               # :154-169
        mean_q = post.mean(0)
        mix = 0.5 * (basis + mean_q)
        log_mix = torch.log(mix + eps)
        kld_p = (0.5 * basis * (torch.log(basis + 1e-12) - log_mix)).sum(dim=1).mean()
        kld_q = (0.5 * mean_q * (torch.log(mean_q + 1e-12) - log_mix)).sum(dim=1).mean()
        return kld_p + kld_q
    raise NotImplementedError(f"Invalid divergence type / Not implemented: {divergence}")


def ls_discriminator_loss(y_real: Tensor, y_fake: Tensor) -> Tensor:
    """LSDiscriminatorialLoss.forward, discriminator/losses.py:11-14."""
    return 0.5 * torch.mean((y_real - 1) ** 2) + 0.5 * torch.mean((y_fake + 1) ** 2)


def ls_generator_loss(y_fake: Tensor) -> Tensor:
    """LSGeneratorLoss.forward, discriminator/losses.py:22-24."""
    return 0.5 * torch.mean((y_fake - 1) ** 2)


# --------------------------------------------------------------------------- #
# discriminator
# --------------------------------------------------------------------------- #
def spectral_weight(P: State, prefix: str, training: bool, eps: float = 1e-12) -> Tensor:
    """torch.nn.utils.spectral_norm (legacy hook form, n_power_iterations=1) as used at
    discriminator/blocks.py:105-108: one power iteration per training-mode forward on
    W = weight_orig.view(Cout, -1), updating weight_u / weight_v in place (no grad);
    sigma = u . (W v) carries grad through W only."""
    w = P[_k(prefix, "weight_orig")]
    u, v = P[_k(prefix, "weight_u")], P[_k(prefix, "weight_v")]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    uu, vv = u.clone(), v.clone()
    sigma = torch.dot(uu, torch.mv(wm, vv))
    return w / sigma


def discriminator_forward(ys: Sequence[Tensor], P: State, prefix: str = "discriminator",
                          depth: int = 4, training: bool = True,
                          noise: Optional[Tensor] = None, flip: Optional[bool] = None,
                          instance_noise: bool = True, label_noise: bool = True,
                          is_training_flag: bool = True) -> Tensor:
    """DiscriminatorBlock.forward, discriminator/blocks.py:114-130.

    `noise` is the (H, W) plane InstanceNoise draws with torch.normal(0, .2) on the CPU
    generator (blocks.py:149-154) and `flip` the outcome of LabelNoise's uniform<0.1 test
    (blocks.py:165-170, utils.py:20-22).  When None they are drawn here in the reference's
    order (normal first, uniform second) from the global CPU generator.
    `training` is nn.Module.training (drives the spectral-norm power iteration);
    `is_training_flag` is the ctor's is_training (drives whether noise is added)."""
    p = prefix + "." if prefix else ""
    s = ys[0]
    if instance_noise:
        if noise is None:
            noise = torch.normal(mean=0.0, std=0.2, size=tuple(s.shape[2:]))
        if is_training_flag:
            s = s + noise.to(s.dtype)
        s = torch.clip(s, 0, 1)                                           # clipping=True :42,152-153
        conv_key = "stack_0.1"
    else:
        conv_key = "stack_0.0"
    s = F.conv2d(s, P[p + conv_key + ".weight"], P[p + conv_key + ".bias"], stride=2, padding=1)
    s = F.leaky_relu(s, 0.2)                                              # :46-51
    for i in range(depth):                                                # :121-125
        s = torch.sigmoid(F.conv2d(s, P[p + f"squeeze_dict.squeeze_{i}.0.weight"],
                                   P[p + f"squeeze_dict.squeeze_{i}.0.bias"]))
        s = torch.cat((s, ys[i + 1]), dim=1)
        w = spectral_weight(P, p + f"spectral_dict.spectral_{i}.0", training)
        s = torch.tanh(F.conv2d(s, w, P[p + f"spectral_dict.spectral_{i}.0.bias"], stride=2, padding=1))
    logits = F.conv2d(s, P[p + "out.0.weight"], P[p + "out.0.bias"]).flatten(1)   # :68-75
    if label_noise:
        if flip is None:
            flip = bool(torch.FloatTensor(1).uniform_(0, 1) < 0.1)
        if flip:
            logits = -1 * logits
    return logits


# --------------------------------------------------------------------------- #
# train step (a17: absent from the reference, models/octa.py:59-60)
# --------------------------------------------------------------------------- #
def mask_pyramid(mask: Tensor, levels: int = 5) -> List[Tensor]:
    """Real multi-scale pyramid for the discriminator: nearest down-sampling by 2**i
    (the contract of discriminator/blocks.py:114-125; built by the absent loop)."""
    out = [mask]
    for i in range(1, levels):
        f = 2 ** i
        out.append(mask[:, :, ::f, ::f].contiguous())
    return out


def segmentor_loss(P: State, x: Tensor, ys: Tensor, *, use_dice: bool = True, kl_weight: float = 0.1,
                   adv_weight: float = 0.1, adversarial: bool = True, noise=None, flip=None,
                   num_classes: int = 2):
    att, agg, _ = resnest_unet_forward(x, P)
    prob = F.softmax(agg, dim=1)
    loss = weighted_partial_ce(prob, ys, num_classes)
    if use_dice:
        loss = loss + dice_loss(prob, ys)
    if adversarial:
        loss = loss + kl_weight * interlayer_divergence([prob, *att])
        d_fake = discriminator_forward(att, P, noise=noise, flip=flip)
        loss = loss + adv_weight * ls_generator_loss(d_fake)
    return loss, att, agg


def discriminator_loss(P: State, real_pyr: Sequence[Tensor], att: Sequence[Tensor],
                       noise_r=None, flip_r=None, noise_f=None, flip_f=None):
    d_real = discriminator_forward(real_pyr, P, noise=noise_r, flip=flip_r)
    d_fake = discriminator_forward([a.detach() for a in att], P, noise=noise_f, flip=flip_f)
    return ls_discriminator_loss(d_real, d_fake)


# --------------------------------------------------------------------------- #
# the rest of the public surface
# --------------------------------------------------------------------------- #
def weighted_partial_ce_ce(y_hat: Tensor, ys: Tensor, full: bool = False) -> Tensor:
    """WeightedPartialCE.forward with manual=False and two classes, segmentor/losses.py:40-45,56-58:
    nn.CrossEntropyLoss on the masked scores with target long(ys[:, 1])."""
    z = y_hat if full else y_hat * ys                                      # :31-32
    z = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])                      # :43
    t = ys[:, 1:].permute(0, 2, 3, 1).reshape(-1).long()                   # :41,45
    return F.cross_entropy(z, t)                                           # :58


def weighted_partial_ce_bce(y_hat: Tensor, ys: Tensor, full: bool = False) -> Tensor:
    """WeightedPartialCE.forward with num_classes == 1 (manual=True), segmentor/losses.py:48-49."""
    z = y_hat if full else y_hat * ys
    return F.binary_cross_entropy_with_logits(z.permute(0, 2, 3, 1).reshape(-1, 1), ys.permute(0, 2, 3, 1).reshape(-1, 1))


def label_noise_label(x: Tensor, flip: bool) -> Tensor:
    """LabelNoise.flip_label, discriminator/blocks.py:172-177."""
    return torch.abs(1.0 - x) if flip else x


def instance_noise(x: Tensor, noise: Tensor, clipping: bool, is_training: bool = True) -> Tensor:
    """InstanceNoise.forward, discriminator/blocks.py:149-154."""
    out = x + noise.to(x.dtype) if is_training else x
    return torch.clip(out, 0, 1) if clipping else out


def classification_predict(x: Tensor, P: State, method: str, mode: str, prefix: str = "segmentor", training: bool = False):
    """ResnestUNet.classification_predict, segmentor/compose.py:201-230 (encoder_gating=False)."""
    p = prefix + "." if prefix else ""
    att, agg, latent = resnest_unet_forward(x, P, prefix, training=training)
    pred = F.softmax(agg, dim=1)                                           # :210
    if mode == "classic":
        emb = F.linear(latent.mean(dim=(2, 3)), P[p + "linear_head_emb.1.weight"], P[p + "linear_head_emb.1.bias"])  # :82-85
    elif mode == "ae-squash":
        emb = pred.mean(dim=(2, 3))                                        # :215
    elif mode == "ae-extract":                                             # :88-98
        e = F.adaptive_avg_pool2d(pred, (32, 32))
        e = F.relu(F.conv2d(e, P[p + "linear_head_dec.1.weight"], P[p + "linear_head_dec.1.bias"]))
        e = batch_norm(e, P, p + "linear_head_dec.3", training)
        e = F.relu(F.conv2d(e, P[p + "linear_head_dec.4.weight"], P[p + "linear_head_dec.4.bias"]))
        e = batch_norm(e, P, p + "linear_head_dec.6", training)
        emb = F.linear(e.mean(dim=(2, 3)), P[p + "linear_head_dec.8.weight"], P[p + "linear_head_dec.8.bias"])
    else:
        raise NotImplementedError
    cls = F.softmax(emb, dim=1) if method == "softmax" else torch.sigmoid(emb)   # :222-225
    return cls, att, pred


def parallel_head_forward(x: Tensor, P: State, gates: bool, gating_level: int = 3, training: bool = True):
    """ResnestUnetParallelHead.forward (segmentor/compose.py:292-346) / ResnestUnetParallelHeadAttentionGate.forward
    (:440-513): returns (attentions, attentions_c, stacked logits)."""
    x_0_0 = stem(x, P, "encoder_0_1_2", training)
    x_0_1 = F.max_pool2d(x_0_0, 3, 2, 1)
    x_1 = encoder_stage(x_0_1, P, "encoder_1", 0, training)
    x_2 = encoder_stage(x_1, P, "encoder_2", 1, training)
    x_3 = encoder_stage(x_2, P, "encoder_3", 2, training)
    pad_h, pad_w = x_3.shape[2] % 2 == 1, x_3.shape[3] % 2 == 1
    if pad_h or pad_w:
        x_3 = F.pad(x_3, (0, int(pad_w), 0, int(pad_h)))
    x_4 = encoder_stage(x_3, P, "encoder_4", 3, training)
    att, att_c = [], []

    def gate(d, name, on, lst):
        if gates and on:
            d, y = attention_gate(d, P, name)
            lst.append(y)
        return d
    d = torch.cat((x_3, upsampling(x_4, P, "upsampling_4")), dim=1)
    d = d[:, :, :d.shape[2] - int(pad_h), :d.shape[3] - int(pad_w)]
    d = gate(resnest_decoder(d, P, "decoder_4", training), "aag_4", gating_level > 3, att)      # :473
    for lvl, skip in ((3, x_2), (2, x_1), (1, x_0_0)):
        d = resnest_decoder(torch.cat((skip, upsampling(d, P, f"upsampling_{lvl}")), dim=1), P, f"decoder_{lvl}", training)
        d = gate(d, f"aag_{lvl}", gating_level >= lvl, att)
    d = gate(resnest_decoder(upsampling(d, P, "upsampling_0"), P, "decoder_0", training), "aag_0", gating_level >= 0, att)
    dc = resnest_decoder(torch.cat((x_0_0, upsampling(x_1, P, "upsampling_1_c")), dim=1), P, "decoder_1_c", training)   # :332-334
    dc = gate(dc, "aag_1_c", gating_level >= 1, att_c)
    dc = gate(resnest_decoder(upsampling(dc, P, "upsampling_0_c"), P, "decoder_0_c", training), "aag_0_c", gating_level >= 0, att_c)
    att.reverse()
    att_c.reverse()
    agg = F.conv2d(d, P["fc.weight"], P["fc.bias"])
    agg_c = F.conv2d(dc, P["fc_c.weight"], P["fc_c.bias"])
    return tuple(att), tuple(att_c), torch.stack((agg, agg_c), 0)
