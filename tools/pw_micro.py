"""Pointwise / mid-size conv layers of the 50^2 .. 13^2 stages: pure kernel time per algo (20 launches captured into one
hipGraph, so the host is out of the picture), against the HBM time of the layer's algorithmic bytes.
Usage: python tools/pw_micro.py [layer ...]   (OCTA_HIP_LIB selects the library build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from tools.conv8_micro import LAYERS

dev = torch.device("cuda:0")
ALGOS = tuple(int(a) for a in os.environ.get("PW_ALGOS", "1,4,5,6,2,3,8,9,10,11").split(","))
REPS = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * REPS) * 1e3


def run(name):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    F_._ALGO_OVERRIDE = 1
    y = F_.raw_conv_fwd(x, w, None, s, p, g, 0)
    dy = torch.randn_like(y)
    flops = 2.0 * B * y.shape[2] * y.shape[3] * Cout * (Cin // g) * k * k
    mb = (x.numel() + y.numel() + w.numel()) * 2 / 1e6
    line = f"{name:11s} {mb:6.1f} MB = {mb / 5.3:5.1f} us @5.3TB/s"
    for kind in ("fwd", "dgrad"):
        line += f" | {kind}:"
        for algo in ALGOS:
            F_._ALGO_OVERRIDE = algo
            fn = (lambda: F_.raw_conv_fwd(x, w, None, s, p, g, 0)) if kind == "fwd" else (lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g))
            t = graph_time(fn)
            line += f" a{algo} {t:5.1f}"
    F_._ALGO_OVERRIDE = 0
    print(line, flush=True)


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in LAYERS] or ["enc1_c1", "enc1_c3", "enc2_c1", "enc2_c3", "enc3_c1", "enc3_c3", "enc3_ds", "enc4_c1", "enc4_c3", "dec4_1x1", "dec3_1x1", "dec2_1x1"]
    for n in names:
        run(n)
