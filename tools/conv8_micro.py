"""fwd / dgrad: 8-wave kernel (algo 2 / 3) against the 4-wave kernels (algo 1) on real layer shapes: results and time.
Usage: python tools/conv8_micro.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_

dev = torch.device("cuda:0")
# name: (B, Cin, H, W, Cout, k, stride, pad, groups)
LAYERS = {
    "dec2_3x3": (16, 512, 100, 100, 256, 3, 1, 1, 1),
    "dec3_3x3": (16, 1024, 50, 50, 512, 3, 1, 1, 1),
    "dec4_3x3": (16, 2048, 25, 25, 1024, 3, 1, 1, 1),
    "dec1_3x3": (16, 128, 200, 200, 64, 3, 1, 1, 1),
    "dec2_splat": (16, 256, 100, 100, 512, 3, 1, 1, 4),
    "dec3_splat": (16, 512, 50, 50, 1024, 3, 1, 1, 4),
    "dec4_splat": (16, 1024, 25, 25, 2048, 3, 1, 1, 4),
    "dec4_1x1": (16, 2048, 25, 25, 1024, 1, 1, 0, 1),
    "dec3_1x1": (16, 1024, 50, 50, 512, 1, 1, 0, 1),
    "dec2_1x1": (16, 512, 100, 100, 256, 1, 1, 0, 1),
    "enc1_c3": (16, 64, 100, 100, 256, 1, 1, 0, 1),
    "enc1_c1": (16, 256, 100, 100, 64, 1, 1, 0, 1),
    "enc2_c1": (16, 512, 50, 50, 128, 1, 1, 0, 1),
    "enc2_splat": (16, 128, 50, 50, 256, 3, 1, 1, 2),
    "enc2_c3": (16, 128, 50, 50, 512, 1, 1, 0, 1),
    "enc3_c1": (16, 1024, 25, 25, 256, 1, 1, 0, 1),
    "enc3_splat": (16, 256, 25, 25, 512, 3, 1, 1, 2),
    "enc3_c3": (16, 256, 25, 25, 1024, 1, 1, 0, 1),
    "enc3_ds": (16, 512, 25, 25, 1024, 1, 1, 0, 1),
    "enc4_c1": (16, 2048, 13, 13, 512, 1, 1, 0, 1),
    "enc4_splat": (16, 512, 13, 13, 1024, 3, 1, 1, 2),
    "enc4_c3": (16, 512, 13, 13, 2048, 1, 1, 0, 1),
    "up4_adj": (16, 1024, 26, 26, 2048, 2, 2, 0, 1),
    "disc2": (16, 15, 100, 100, 256, 4, 2, 1, 1),
    "s_odd": (3, 40, 13, 11, 136, 3, 1, 1, 1),
    "s_odd2": (2, 24, 9, 10, 72, 3, 2, 1, 1),
}


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(name):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    bias = torch.randn(Cout, device=dev)
    F_._ALGO_OVERRIDE = 1
    y1 = F_.raw_conv_fwd(x, w, bias, s, p, g, 1)
    dy = torch.randn_like(y1)
    dx1 = F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g)
    flops = 2.0 * B * y1.shape[2] * y1.shape[3] * Cout * (Cin // g) * k * k
    ok = True
    line = f"{name:11s}"
    for kind in ("fwd", "dgrad"):
        res = {}
        for algo in (1, 5, 2, 3, 8):
            F_._ALGO_OVERRIDE = algo
            fn = (lambda: F_.raw_conv_fwd(x, w, bias, s, p, g, 1)) if kind == "fwd" else (lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g))
            out = fn()
            ref = y1 if kind == "fwd" else dx1
            err = (out.float() - ref.float()).abs().max().item()
            sc = ref.float().abs().max().item()
            if err > 2e-2 * sc + 1e-6:
                ok = False
                line += f" [{kind} algo{algo} MISMATCH {err:.3e}/{sc:.3e}]"
            res[algo] = timeit(fn)
        line += f" | {kind}: " + " ".join(f"a{a} {t:7.1f}us {flops / t / 1e6:6.1f}TF" for a, t in res.items())
    F_._ALGO_OVERRIDE = 0
    print(line, flush=True)
    return ok


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in LAYERS] or list(LAYERS)
    good = all([run(n) for n in names])
    print("ALL OK" if good else "FAILURES")
    sys.exit(0 if good else 1)
