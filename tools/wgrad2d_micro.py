"""wgrad2d (2-D patch weight gradient, octa_tuning_set(10, 1)) against wgrad9 on single 3x3 layers: one launch behind a 512 MB sweep, alternating.
Usage: python tools/wgrad2d_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd._lib import lib
from tools.wgrad_sched import C, make, job_array, st

dev = torch.device("cuda:0")
L = lib()
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
C["e3_splat_nb"] = (16, 256, 25, 25, 512, 3, 1, 1, 2, 0)
C["d3_splat_nb"] = (16, 512, 50, 50, 1024, 3, 1, 1, 4, 0)
for name in ("d2_3x3", "d3_3x3", "d4_3x3", "d3_splat_nb", "e3_splat_nb"):
    it = make(name)
    dw = torch.zeros_like(it["w"])
    arr = job_array([it], [dw], [None])
    res = {0: [], 1: []}
    kn = {}
    for rnd in range(5):
        for m in (0, 1):
            L.octa_tuning_set(10, m)
            evict.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
            e1.record(); e1.synchronize()
            res[m].append(e0.elapsed_time(e1) * 1e3)
            kn[m] = L.octa_last_conv_kernel().decode()
    L.octa_tuning_set(10, 2)
    out = []
    for m in (0, 1):
        v = sorted(res[m])
        out.append(f"{kn[m]:32s} median {v[2]:7.1f} us min {v[0]:7.1f} ({it['flops'] / v[2] / 1e6:6.1f} TF/s)")
    print(f"{name:12s} | " + " | ".join(out), flush=True)
