"""Tall-skinny weight gradients (few output channels / few input channels, huge pixel count): time of octa_conv2d_wgrad per layer.
Usage: OCTA_WGRAD_BLOCKS=<n> python tools/wgrad_small_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_

dev = torch.device("cuda:0")
# name: (B, Cin, H, W, Cout, k, stride, pad, groups)
LAYERS = {
    "disc0 2->64 k4s2": (16, 2, 400, 400, 64, 4, 2, 1, 1),
    "stem 3->32 k3s2": (16, 3, 400, 400, 32, 3, 2, 1, 1),
    "splat 64->128 g4": (16, 64, 200, 200, 128, 3, 1, 1, 4),
    "splat 64->128 g2": (16, 64, 100, 100, 128, 3, 1, 1, 2),
    "dec0 64->32 k1": (16, 64, 400, 400, 32, 1, 1, 0, 1),
    "head 64->13": (16, 64, 200, 200, 13, 1, 1, 0, 1),
    "dec0 32->32 k3": (16, 32, 200, 200, 32, 3, 1, 1, 1),
}


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


line = f"blocks {os.environ.get('OCTA_WGRAD_BLOCKS', '512'):>5s}:"
for name, (B, Cin, H, W, Cout, k, s, p, g) in LAYERS.items():
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    if Cin % 8:
        x = F_.to_nhwc(x.float()[:, :Cin].contiguous(), dtype=torch.bfloat16)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = F_.nhwc_empty(B, Cout, OH, OW, torch.bfloat16, dev, zero=True); dy.normal_()
    w = torch.randn(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros_like(w)
    t = timeit(lambda: F_.raw_conv_wgrad(x, dy, w, s, p, g, dw=dw))
    line += f" {name} {t:6.1f}us |"
print(line, flush=True)
