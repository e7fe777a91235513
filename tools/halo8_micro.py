"""3x3 layers through the 8-wave kernels: conv_igemm8 (algo 2 / 3), the 4-wave halo kernel (1) and the 2-D patch kernel halo8
(12 / 13); warm, 10 launches per timing, fwd and dgrad.  Usage: python tools/halo8_micro.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib
from tools.conv8_micro import LAYERS, timeit

dev = torch.device("cuda:0")
ALGOS = tuple(int(a) for a in os.environ.get("H8_ALGOS", "1,2,3,12").split(","))


def run(name):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    bias = torch.randn(Cout, device=dev)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    F_.set_splitk_workspace(ws)
    F_._ALGO_OVERRIDE = 1
    y1 = F_.raw_conv_fwd(x, w, bias, s, p, g, 1)
    dy = torch.randn_like(y1)
    dx1 = F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g)
    flops = 2.0 * B * y1.shape[2] * y1.shape[3] * Cout * (Cin // g) * k * k
    line = f"{name:11s}"
    for kind in ("fwd", "dgrad"):
        line += f" | {kind}:"
        for algo in ALGOS:
            # 12 = halo8 with the packed (bank-conflict-free) patch image, 112 = the same kernel with the linear image of round 4
            lib().octa_tuning_set(6, 0 if algo == 112 else 1)
            algo = 12 if algo == 112 else algo
            F_._ALGO_OVERRIDE = algo
            fn = (lambda: F_.raw_conv_fwd(x, w, bias, s, p, g, 1)) if kind == "fwd" else (lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g))
            out = fn()
            kn = lib().octa_last_conv_kernel().decode()
            ref = y1 if kind == "fwd" else dx1
            err = (out.float() - ref.float()).abs().max().item()
            sc = ref.float().abs().max().item()
            bad = " MISMATCH" if err > 2e-2 * sc + 1e-6 else ""
            t = timeit(fn)
            line += f" a{algo}[{kn.split('<')[0].replace('conv_', '').replace('_kernel', '')}{kn[kn.find(','):-1] if ',' in kn else ''}] {t:6.1f}us {flops / t / 1e6:6.1f}TF{bad}"
    F_._ALGO_OVERRIDE = 0
    lib().octa_tuning_set(6, 1)
    F_.set_splitk_workspace(None)
    print(line, flush=True)


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in LAYERS] or ["dec2_3x3", "dec3_3x3", "dec4_3x3", "dec1_3x3", "dec2_splat", "dec3_splat", "dec4_splat", "enc2_splat", "enc3_splat", "enc4_splat"]
    for n in names:
        run(n)
