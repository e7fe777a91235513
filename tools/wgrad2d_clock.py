"""In-kernel clock (s_memtime / s_memrealtime stamps, DIAGNOSTIC build) of wgrad9 and wgrad2d on the decoder 3x3 weight gradients: one launch
behind a 512 MB sweep and one behind ~1 s of back-to-back launches of itself.  Usage (GPU box):
    bash octave_amd/csrc/build.sh diag && OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so python tools/wgrad2d_clock.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from octave_amd._lib import lib
from tools.wgrad_sched import make, job_array, st

assert "diag" in os.environ.get("OCTA_HIP_LIB", ""), "run with OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so"
L = lib()
dll = L._dll
evict = torch.empty(512 << 20, dtype=torch.uint8, device="cuda:0")


def read():
    buf = (ctypes.c_uint64 * (4096 * 4))()
    assert dll.octa_diag_stamps_read_wgrad9(buf, 0) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.float64)
    return a[a[:, 3] > a[:, 1]]


for name in sys.argv[1:] or ("d2_3x3", "d3_3x3", "d4_3x3"):
    it = make(name)
    dw = torch.zeros_like(it["w"])
    arr = job_array([it], [dw], [None])
    for m in (0, 1):
        L.octa_tuning_set(10, m)
        fn = lambda: L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
        for warm in (0, 1):
            if warm:
                t0 = time.time()
                while time.time() - t0 < 1.0:
                    for _ in range(20):
                        fn()
                    torch.cuda.synchronize()
            else:
                fn(); torch.cuda.synchronize(); evict.zero_()
            torch.cuda.synchronize()
            dll.octa_diag_stamps_read_wgrad9(None, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            a = read()
            cyc, real = a[:, 2] - a[:, 0], (a[:, 3] - a[:, 1]) * 10e-9
            clk = cyc / real / 1e9
            span = (a[:, 3].max() - a[:, 1].min()) * 10e-3
            print(f"{name:8s} {L.octa_last_conv_kernel().decode():32s} {'back-to-back' if warm else 'behind sweep':12s} launch {us:6.1f} us | {len(a):4d} workgroups, main loop median {np.median(real) * 1e6:6.1f} us "
                  f"({np.median(cyc) / 1e3:6.1f} k cycles), first begin -> last end {span:6.1f} us | clock median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f})", flush=True)
    L.octa_tuning_set(10, 2)
