"""The statistics pass (octa_bn_stats = bn_reduce_kernel<.,0> + bn_stats_finalize_kernel) and the backward reduction on the big
BatchNorm shapes of the training step, one call behind a 512 MB sweep (cold) and the same call repeated (warm), bf16.
Usage: python tools/bn_reduce_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()
SHAPES = [(16, 32, 400, 400), (16, 32, 200, 200), (16, 64, 200, 200), (16, 128, 200, 200), (16, 64, 100, 100), (16, 128, 100, 100), (16, 256, 100, 100),
          (16, 512, 50, 50), (16, 1024, 25, 25)]
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def ev(fn, cold):
    ts = []
    for _ in range(7):
        if cold:
            evict.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[3]


print(f"{'shape':>22s} {'MB':>6s} | stats cold us  TB/s | warm us  TB/s | bwd-reduce+apply cold us  TB/s(5 passes)")
for B, C, H, W in SHAPES:
    t = F_.nhwc_empty(B, C, H, W, torch.bfloat16, dev, zero=True); t.normal_()
    rows = B * H * W
    mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
    ws = F_._bn_ws(rows, C, dev)
    st = torch.cuda.current_stream().cuda_stream

    def stats():
        L.octa_bn_stats(t.data_ptr(), rows, C, F_.nhwc_ld(t), 0, F_._dt(t), 1e-5, 0.1, mean.data_ptr(), invstd.data_ptr(), None, None, ws.data_ptr(), st)
    mb = t.numel() * 2 / 1e6
    c, w = ev(stats, True), ev(stats, False)
    g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y, mean2, invstd2, xs, mask = F_.raw_bn_fwd(t, g, b_, rm, rv, 0.1, 1e-5, True, True)
    dy = torch.randn_like(t)
    tb = ev(lambda: F_.raw_bn_bwd(dy, xs, y, mean2, invstd2, g, True, False, dg, db, mask), True)
    print(f"{str((B, C, H, W)):>22s} {mb:6.1f} | {c:8.1f} {mb / c:6.2f} | {w:7.1f} {mb / w:5.2f} | {tb:8.1f} {5 * mb / tb:5.2f}", flush=True)
