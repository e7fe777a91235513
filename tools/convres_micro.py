"""Resident-weight persistent conv kernel (algo 7) against the heuristic kernels (algo 1) on the wide shallow layers of the step,
plus ablations (OCTA_CONVRES_DBG: 1 no stores, 2 no MFMA loop, 4 no patch DMA).  Usage: python tools/convres_micro.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_

dev = torch.device("cuda:0")
# name: (B, Cin, H, W, Cout, k, pad)
LAYERS = {
    "dec0_3x3a": (16, 64, 400, 400, 32, 3, 1),
    "dec0_3x3b": (16, 32, 400, 400, 64, 3, 1),
    "dec0_1x1": (16, 64, 400, 400, 32, 1, 0),
    "head13": (16, 64, 200, 200, 13, 1, 0),
    "up1": (16, 64, 200, 200, 256, 1, 0),
    "enc1_c3": (16, 64, 100, 100, 256, 1, 0),
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
    B, Cin, H, W, Cout, k, p = LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    bias = torch.randn(Cout, device=dev)
    F_._ALGO_OVERRIDE = 1
    y1 = F_.raw_conv_fwd(x, w, bias, 1, p, 1, 1)
    dy = torch.randn_like(y1)
    gb = (x.numel() + y1.numel()) * 2 / 1e9
    line = f"{name:10s} {gb * 1e3:6.0f} MB"
    for kind in ("fwd", "dgrad"):
        if kind == "dgrad" and Cout not in (32, 64):
            continue
        fn = (lambda: F_.raw_conv_fwd(x, w, bias, 1, p, 1, 1)) if kind == "fwd" else (lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), 1, p, 1))
        line += f" | {kind}:"
        for algo, dbg in ((1, 0), (7, 0), (7, 16), (7, 2), (7, 18), (7, 1)):
            F_._ALGO_OVERRIDE = algo
            os.environ["OCTA_CONVRES_DBG"] = str(dbg)
            t = timeit(fn)
            line += f" a{algo}{'/d%d' % dbg if dbg else ''} {t:6.1f}us ({gb / t * 1e3:4.2f}TB/s)" if dbg == 0 else f" d{dbg} {t:6.1f}"
    os.environ["OCTA_CONVRES_DBG"] = "0"
    F_._ALGO_OVERRIDE = 0
    print(line, flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(LAYERS)):
        run(n)
