"""BatchNorm kernels on the BN shapes of the training step: time and achieved bandwidth of forward (stats + apply) and backward
(reduce + finalize + apply).  Usage: python tools/bn_micro.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from architectures.models.octa import OctaScribbleNet
from octave_amd import functional as F_
from octave_amd import layers as L_

B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
shapes = collections.Counter()
hooks = []
for m in net.modules():
    if isinstance(m, L_.BatchNorm2d):
        hooks.append(m.register_forward_pre_hook(lambda mod, args: shapes.update([tuple(args[0].shape)])))
x = torch.randn(B, 3, H, H, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=False):
    try:
        net.segmentor.compute_dtype = torch.bfloat16
    except Exception:
        pass
    net.segmentor(F_.to_nhwc(x, dtype=torch.bfloat16))
for h in hooks:
    h.remove()


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = collections.Counter()
print(f"{'shape':>24s} {'n':>3s} {'MB':>6s} | {'fwd us':>8s} {'TB/s':>5s} | {'bwd us':>8s} {'TB/s':>5s}")
for shp, n in sorted(shapes.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] * kv[1]):
    Bb, C, Hh, Ww = shp
    t = F_.nhwc_empty(Bb, C, Hh, Ww, torch.bfloat16, dev, zero=True); t.normal_()
    dy = torch.randn_like(t)
    g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y, mean, invstd, xs, mask = F_.raw_bn_fwd(t, g, b_, rm, rv, 0.1, 1e-5, True, True)
    mb = t.numel() * 2 / 1e6
    tf = timeit(lambda: F_.raw_bn_fwd(t, g, b_, rm, rv, 0.1, 1e-5, True, True))
    tb = timeit(lambda: F_.raw_bn_bwd(dy, xs, y, mean, invstd, g, True, False, dg, db, mask))
    print(f"{str(shp):>24s} {n:3d} {mb:6.1f} | {tf:8.1f} {3 * mb / tf:5.2f} | {tb:8.1f} {5 * mb / tb:5.2f}")
    tot["fwd"] += tf * n; tot["bwd"] += tb * n; tot["mb"] += mb * n
print(f"sum over the step: fwd {tot['fwd'] / 1e3:.2f} ms (3 passes of {tot['mb'] / 1e3:.2f} GB = {3 * tot['mb'] / tot['fwd']:.2f} TB/s), "
      f"bwd {tot['bwd'] / 1e3:.2f} ms (5 passes = {5 * tot['mb'] / tot['bwd']:.2f} TB/s)")
