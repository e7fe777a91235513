"""Micro-benchmark of single conv-engine launches (fwd / dgrad / wgrad) on real layer shapes.
Usage: python tools/conv_micro.py [layer ...] ; env OCTA_CONV_VARIANT selects the igemm variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_

LAYERS = {
    # name: (B, Cin, H, W, Cout, k, stride, pad, groups)
    "dec2_3x3": (16, 512, 100, 100, 256, 3, 1, 1, 1),
    "dec3_3x3": (16, 1024, 50, 50, 512, 3, 1, 1, 1),
    "dec4_3x3": (16, 2048, 25, 25, 1024, 3, 1, 1, 1),
    "dec1_3x3": (16, 128, 200, 200, 64, 3, 1, 1, 1),
    "dec0_3x3": (16, 64, 400, 400, 32, 3, 1, 1, 1),
    "enc1_1x1": (16, 256, 100, 100, 64, 1, 1, 0, 1),
    "enc3_1x1": (16, 256, 25, 25, 1024, 1, 1, 0, 1),
    "enc2_splat": (16, 128, 50, 50, 256, 3, 1, 1, 2),
    "dec0_splat": (16, 32, 400, 400, 64, 3, 1, 1, 4),
    "dec2_splat": (16, 256, 100, 100, 512, 3, 1, 1, 4),
    "disc0": (16, 2, 400, 400, 64, 4, 2, 1, 1),
    "disc1": (16, 15, 200, 200, 128, 4, 2, 1, 1),
    "disc2": (16, 141, 100, 100, 256, 4, 2, 1, 1),
    "dec0_1x1": (16, 64, 400, 400, 32, 1, 1, 0, 1),
    "up1": (16, 64, 200, 200, 256, 1, 1, 0, 1),
}


def run(name, reps=10, kinds=("fwd", "dgrad", "wgrad")):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
    dev = torch.device("cuda:0")
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev)
    x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    y = F_.raw_conv_fwd(x, w, None, s, p, g)
    dy = torch.randn_like(y)
    flops = 2.0 * B * y.shape[2] * y.shape[3] * Cout * (Cin // g) * k * k
    for kind in kinds:
        fn = {"fwd": lambda: F_.raw_conv_fwd(x, w, None, s, p, g),
              "dgrad": lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g),
              "wgrad": lambda: F_.raw_conv_wgrad(x, dy, w, s, p, g)}[kind]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"{name:12s} {kind:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  ({flops / 1e9:.1f} GFLOP)", flush=True)


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in LAYERS] or list(LAYERS)
    kinds = tuple(a for a in sys.argv[1:] if a in ("fwd", "dgrad", "wgrad")) or ("fwd", "dgrad", "wgrad")
    for n in names:
        run(n, kinds=kinds)
