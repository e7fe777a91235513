"""In-kernel clock of the three MFMA-bound kernels (MI355X_MICROARCH.md, DVFS give-back item 6): the DIAGNOSTIC build of the library
(octave_amd/csrc/build.sh diag -> libocta_hip_diag.so, stamps of s_memtime / s_memrealtime around each workgroup's main loop) runs
every kernel back to back on random data for ~2 s, then one more launch is read out: clock = d memtime / d memrealtime x 100 MHz,
median over the workgroups, beside the main loop's share of the launch.  Usage (GPU box):
    OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so python tools/clock_stamps.py > profiles/r05_clock_stamps.txt"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib, WgradJob
from tools.conv8_micro import LAYERS

assert "diag" in os.environ.get("OCTA_HIP_LIB", ""), "run with OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so"
dev = torch.device("cuda:0")
L = lib()
dll = L._dll


def read(which):
    buf = (ctypes.c_uint64 * (4096 * 4))()
    if which == "wgrad9":
        rc = dll.octa_diag_stamps_read_wgrad9(buf, 0)
    else:
        rc = dll.octa_diag_stamps_read_conv(0 if which == "halo" else 1, buf, 0)
    assert rc == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.float64)
    a = a[a[:, 3] > a[:, 1]]
    return a


def clear(which):
    if which == "wgrad9":
        dll.octa_diag_stamps_read_wgrad9(None, 1)
    else:
        dll.octa_diag_stamps_read_conv(0 if which == "halo" else 1, None, 1)


def run(name, which, fn, secs=2.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    clear(which)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    a = read(which)
    cyc, real = a[:, 2] - a[:, 0], (a[:, 3] - a[:, 1]) * 10e-9
    clk = cyc / real / 1e9
    print(f"{name:34s} {L.octa_last_conv_kernel().decode():40s} launch {us:7.1f} us | workgroups stamped {len(a):4d} | main loop {np.median(real) * 1e6:6.1f} us (median), "
          f"{np.median(cyc) / 1e3:7.1f} k cycles | in-kernel clock median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f})", flush=True)


def conv_case(layer, algo, kind):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[layer]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    F_._ALGO_OVERRIDE = 1
    y = F_.raw_conv_fwd(x, w, None, s, p, g, 0)
    dy = torch.randn_like(y)
    def fn():
        F_._ALGO_OVERRIDE = algo
        if kind == "fwd":
            F_.raw_conv_fwd(x, w, None, s, p, g, 0)
        else:
            F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g)
        F_._ALGO_OVERRIDE = 0
    return fn, (x, w, dy)


def wgrad_case(layer):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[layer]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = F_.nhwc_empty(B, Cout, OH, OW, torch.bfloat16, dev, zero=True); dy.normal_()
    dw = torch.zeros(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    d = F_._desc(B, H, W, OH, OW, Cin, Cout, k, k, s, p, g, F_.nhwc_ld(x), F_.nhwc_ld(dy), torch.bfloat16)
    jobs = (WgradJob * 1)()
    ctypes.memmove(ctypes.byref(jobs[0].d), ctypes.byref(d), ctypes.sizeof(d))
    jobs[0].x, jobs[0].dy, jobs[0].dw, jobs[0].dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None
    for a in range(4):
        jobs[0].dw_strides[a] = dw.stride(a)
    st = torch.cuda.current_stream().cuda_stream
    return (lambda: L.octa_conv2d_wgrad_batch(jobs, 1, None, 0, st)), (x, dy, dw, jobs)


if __name__ == "__main__":
    print("in-kernel clock stamps (diagnostic build; random bf16 operands; every kernel behind ~2 s of back-to-back launches of itself)")
    for layer in ("dec2_3x3", "dec3_3x3"):
        for algo, which, tag in ((12, "halo", "halo8 32x32x16"), (13, "halo", "halo16 16x16x32"), (14, "halo", "halo16p persistent"), (3, "igemm8", "igemm8 128x256 16x16x32")):
            for kind in ("fwd", "dgrad"):
                fn, keep = conv_case(layer, algo, kind)
                run(f"{layer} {kind} {tag}", which, fn)
        fn, keep = wgrad_case(layer)
        run(f"{layer} wgrad9 32x32x16", "wgrad9", fn)
