"""Per-pass byte ledger of the BatchNorm and split-attention launches of one training step (BASELINE configs[2]: B = 16, 400 x 400,
bf16): for every call, the bytes its passes move over the activation tensors (as implemented: octa_bn_train_fwd = statistics
pass + apply pass, ...), the minimum any implementation needs (every input tensor read once, every output written once, per
direction), the HIP-event time of the call in an eagerly launched step and the rate that implies.  The sum says how much of the
BatchNorm + split-attention time is streaming at the copy rate and how much is passes that a fusion could remove.
Usage (GPU box): python tools/bn_ledger.py [B] [H] > profiles/rNN_bn_splat_ledger.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from octave_amd import functional as F_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
LEDGER = []          # (kind, label, moved bytes, minimum bytes, event0, event1)


def timed(kind, label, moved, minimum, fn, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn(*a, **kw)
    e1.record()
    LEDGER.append((kind, label, moved, minimum, e0, e1))
    return out


_fwd, _bwd = F_.raw_bn_fwd, F_.raw_bn_bwd


def bn_fwd(x, gamma, beta, rm, rv, momentum, eps, training, relu, residual=None, pre_sums=None):
    Bn, C, Hh, Ww = x.shape
    t = Bn * C * Hh * Ww * x.element_size()
    res = residual is not None
    # as implemented: statistics pass (1 read of x) + apply pass (1 read of x [+ 1 read of the residual], 1 write of y) + a 1-bit mask
    moved = t * (1 + 1 + (1 if res else 0) + 1) + (t // 16 if relu else 0)
    # minimum: x (and the residual) read once, y written once -- what a statistics-in-the-producer + apply-in-the-consumer design would move is 0
    minimum = t * (1 + (1 if res else 0) + 1)
    return timed("bn fwd", f"C={C} {Hh}x{Ww}{' +res' if res else ''}{' relu' if relu else ''}", moved, minimum, _fwd, x, gamma, beta, rm, rv, momentum, eps, training, relu, residual, pre_sums)


def bn_bwd(dy, x, y, mean, invstd, gamma, relu, want_dres, dgamma, dbeta, mask=None):
    Bn, C, Hh, Ww = x.shape
    t = Bn * C * Hh * Ww * x.element_size()
    m = t // 16 if mask is not None else (t if (relu and y is not None) else 0)
    # as implemented: reduction pass (dy, x [, mask / y]) + apply pass (dy, x [, mask / y]; writes dx [and the residual's gradient])
    moved = (2 * t + m) + (2 * t + m) + t * (1 + (1 if want_dres else 0))
    minimum = 2 * t + m + t * (1 + (1 if want_dres else 0))
    return timed("bn bwd", f"C={C} {Hh}x{Ww}{' +dres' if want_dres else ''}{' relu' if relu else ''}", moved, minimum, _bwd, dy, x, y, mean, invstd, gamma, relu, want_dres, dgamma, dbeta, mask)


F_.raw_bn_fwd, F_.raw_bn_bwd = bn_fwd, bn_bwd

_sf, _sb = F_.SplatTailFn.forward, F_.SplatTailFn.backward


def splat_fwd(ctx, xr, *a, **kw):
    Bn, C2, Hh, Ww = xr.shape
    t2 = Bn * C2 * Hh * Ww * xr.element_size()
    # as implemented (bn0 on the fly): statistics (1 read of the 2C tensor), GAP (1 read), weighted sum (1 read, 1 write of C)
    moved, minimum = 3 * t2 + t2 // 2, t2 + t2 // 2
    return timed("splat fwd", f"2C={C2} {Hh}x{Ww}", moved, minimum, _sf, ctx, xr, *a, **kw)


def splat_bwd(ctx, dout):
    Bn, C, Hh, Ww = dout.shape
    t = Bn * C * Hh * Ww * dout.element_size()
    t2 = 2 * t
    # as implemented: logit gradients (dout, x, out) [+ bn0's sums in a pass of their own with OCTA_SPLAT_BWD_MERGE=0], dx (dout, x, out;
    # writes the 2C gradient)
    passes = 2 if F_._SPLAT_BWD_MERGE else 3
    moved, minimum = passes * (t + t2 + t) + t2, (t + t2 + t) + t2
    return timed("splat bwd", f"C={C} {Hh}x{Ww}", moved, minimum, _sb, ctx, dout)


F_.SplatTailFn.forward = staticmethod(splat_fwd)
F_.SplatTailFn.backward = staticmethod(splat_bwd)

from architectures.models.octa import OctaScribbleNet  # noqa: E402
from octave_amd.train import TrainStep, mask_pyramid  # noqa: E402

torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
x, ys, real = F_.synth_octa_batch(B, H, H, seed=1234, device=dev)
batch = (x, ys, mask_pyramid(real))
for _ in range(2):
    step(*batch)
torch.cuda.synchronize()
LEDGER.clear()
step(*batch)
torch.cuda.synchronize()
step.close()

agg = collections.OrderedDict()
rows = []
for kind, label, moved, minimum, e0, e1 in LEDGER:
    ms = e0.elapsed_time(e1)
    a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += moved; a[2] += minimum; a[3] += ms
    rows.append((kind, label, moved, minimum, ms))
print(f"BatchNorm / split-attention byte ledger, one eagerly launched step, B = {B}, {H} x {H}, bf16 (times are HIP events around each call in an EAGER step:")
print("they include the launch gaps of the 2-3 kernels of a call; the replayed step's kernel times are in the *_stats_summary.txt of the round)")
print(f"{'kind':10s} {'calls':>5s} {'moved MB':>10s} {'minimum MB':>11s} {'moved/min':>9s} {'ms':>8s} {'TB/s moved':>10s}")
tot = [0, 0.0, 0.0, 0.0]
for k, (n, mv, mn, ms) in agg.items():
    print(f"{k:10s} {n:5d} {mv / 1e6:10.1f} {mn / 1e6:11.1f} {mv / mn:9.2f} {ms:8.3f} {mv / ms / 1e9:10.2f}")
    for i, v in enumerate((n, mv, mn, ms)):
        tot[i] += v
print(f"{'total':10s} {tot[0]:5d} {tot[1] / 1e6:10.1f} {tot[2] / 1e6:11.1f} {tot[1] / tot[2]:9.2f} {tot[3]:8.3f} {tot[1] / tot[3] / 1e9:10.2f}")
print(f"at the copy rate of this part (6.3 TB/s, MI355X_MICROARCH.md) the moved bytes take {tot[1] / 6.3e9:.2f} ms, the minimum {tot[2] / 6.3e9:.2f} ms")
print()
print("the 40 largest calls:")
for kind, label, moved, minimum, ms in sorted(rows, key=lambda r: -r[4])[:40]:
    print(f"  {kind:10s} {label:28s} moved {moved / 1e6:8.1f} MB  minimum {minimum / 1e6:8.1f} MB  {ms * 1e3:8.1f} us  {moved / ms / 1e9:6.2f} TB/s")
