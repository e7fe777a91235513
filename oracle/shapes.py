"""Shapes of the reference's ``OctaScribbleNet.state_dict()`` derived from its constructors
(test infrastructure; lets the oracle build a state without importing the reference).

Follows models/octa.py:44-49, segmentor/compose.py:24-98, extra/resnest.py:298-366 &
376-429 (resnest50: layers [3,4,6,3], stem 32, radix 2, cardinality 1),
discriminator/blocks.py:36-79.
"""
from collections import OrderedDict


def _bn(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)
    d[p + ".running_mean"] = (c,)
    d[p + ".running_var"] = (c,)
    d[p + ".num_batches_tracked"] = ()


def _splat(d, p, cin, ch, card, bias):
    inter = max(cin * 2 // 4, 32)                      # resnest.py:76
    d[p + ".conv.weight"] = (ch * 2, cin // (card * 2), 3, 3)
    if bias:
        d[p + ".conv.bias"] = (ch * 2,)
    _bn(d, p + ".bn0", ch * 2)
    d[p + ".fc1.weight"] = (inter, ch // card, 1, 1)
    d[p + ".fc1.bias"] = (inter,)
    _bn(d, p + ".bn1", inter)
    d[p + ".fc2.weight"] = (ch * 2, inter // card, 1, 1)
    d[p + ".fc2.bias"] = (ch * 2,)


def _bottleneck(d, p, inpl, planes, down):
    d[p + ".conv1.weight"] = (planes, inpl, 1, 1)
    _bn(d, p + ".bn1", planes)
    _splat(d, p + ".conv2", planes, planes, 1, False)
    d[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
    _bn(d, p + ".bn3", planes * 4)
    if down:
        d[p + ".downsample.1.weight"] = (planes * 4, inpl, 1, 1)
        _bn(d, p + ".downsample.2", planes * 4)


def _decoder(d, p, cin, cout):
    d[p + ".conv.0.weight"] = (cout, cin, 3, 3)
    _bn(d, p + ".conv.1", cout)
    _splat(d, p + ".conv.3", cout, cout, 2, True)
    d[p + ".downsample.0.weight"] = (cout, cin, 1, 1)
    _bn(d, p + ".downsample.1", cout)


def segmentor_shapes(num_classes=2, prefix="segmentor"):
    d = OrderedDict()
    p = prefix + "." if prefix else ""
    s = p + "encoder_0_1_2"
    d[s + ".0.0.weight"] = (32, 3, 3, 3)
    _bn(d, s + ".0.1", 32)
    d[s + ".0.3.weight"] = (32, 32, 3, 3)
    _bn(d, s + ".0.4", 32)
    d[s + ".0.6.weight"] = (64, 32, 3, 3)
    _bn(d, s + ".1", 64)
    inpl = 64
    for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for b in range(blocks):
            _bottleneck(d, f"{p}encoder_{i + 1}.{b}", inpl, planes, b == 0)
            inpl = planes * 4
    for lvl, (ui, uo, di, do) in enumerate(((64, 64, 64, 32), (256, 64, 128, 64), (512, 256, 512, 256),
                                            (1024, 512, 1024, 512), (2048, 1024, 2048, 1024))):
        d[f"{p}upsampling_{lvl}.up.weight"] = (ui, uo, 2, 2)
        d[f"{p}upsampling_{lvl}.up.bias"] = (uo,)
        _decoder(d, f"{p}decoder_{lvl}", di, do)
        d[f"{p}aag_{lvl}.conv1.weight"] = (num_classes, do, 1, 1)
        d[f"{p}aag_{lvl}.conv1.bias"] = (num_classes,)
    d[p + "fc.weight"] = (num_classes, 32, 1, 1)
    d[p + "fc.bias"] = (num_classes,)
    d[p + "linear_head_emb.1.weight"] = (num_classes, 2048)
    d[p + "linear_head_emb.1.bias"] = (num_classes,)
    d[p + "linear_head_dec.1.weight"] = (64, num_classes, 7, 7)
    d[p + "linear_head_dec.1.bias"] = (64,)
    _bn(d, p + "linear_head_dec.3", 64)
    d[p + "linear_head_dec.4.weight"] = (512, 64, 7, 7)
    d[p + "linear_head_dec.4.bias"] = (512,)
    _bn(d, p + "linear_head_dec.6", 512)
    d[p + "linear_head_dec.8.weight"] = (num_classes, 512)
    d[p + "linear_head_dec.8.bias"] = (num_classes,)
    return d


def discriminator_shapes(H, W=None, in_ch=2, nf=64, depth=4, prefix="discriminator"):
    W = H if W is None else W
    d = OrderedDict()
    p = prefix + "." if prefix else ""
    d[p + "stack_0.1.weight"] = (nf, in_ch, 4, 4)
    d[p + "stack_0.1.bias"] = (nf,)
    for i in range(depth):
        d[f"{p}squeeze_dict.squeeze_{i}.0.weight"] = (13, nf * 2 ** i, 1, 1)
        d[f"{p}squeeze_dict.squeeze_{i}.0.bias"] = (13,)
    for i in range(depth):
        co = nf * 2 * 2 ** i
        d[f"{p}spectral_dict.spectral_{i}.0.bias"] = (co,)
        d[f"{p}spectral_dict.spectral_{i}.0.weight_orig"] = (co, 13 + in_ch, 4, 4)
        d[f"{p}spectral_dict.spectral_{i}.0.weight_u"] = (co,)
        d[f"{p}spectral_dict.spectral_{i}.0.weight_v"] = ((13 + in_ch) * 16,)
    d[p + "out.0.weight"] = (1, nf * 2 ** depth, H // 2 ** (depth + 1), W // 2 ** (depth + 1))
    d[p + "out.0.bias"] = (1,)
    return d


def octa_state_shapes(H, with_disc=True, num_classes=2):
    d = segmentor_shapes(num_classes)
    if with_disc:
        d.update(discriminator_shapes(H, in_ch=num_classes))
    return d
