"""Which Python call sites still zero-fill tensors during one training step?  (wraps torch.zeros / zeros_like / ones_like /
Tensor.zero_ / Tensor.fill_ and counts callers)  usage: python tools/fill_sites.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid

B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, compute_dtype=torch.bfloat16)
x, ys, real = bench.synth_batch(B, H, 0, dev)
batch = (x, ys, mask_pyramid(real))
step(*batch); step(*batch)
torch.cuda.synchronize()
sites = collections.Counter()
bytes_ = collections.Counter()


def where():
    st = [f for f in traceback.extract_stack()[:-2] if "site-packages" not in f.filename and "fill_sites" not in f.filename]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))


def wrap(mod, name, is_method=False):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        t = out if isinstance(out, torch.Tensor) else (a[0] if a and isinstance(a[0], torch.Tensor) else None)
        if t is not None and t.is_cuda:
            w = f"{name:10s} " + where()
            sites[w] += 1
            bytes_[w] += t.numel() * t.element_size()
        return out
    setattr(mod, name, f)
    return orig


saved = [(torch, n, wrap(torch, n)) for n in ("zeros", "zeros_like", "ones_like", "ones", "full")]
saved += [(torch.Tensor, n, wrap(torch.Tensor, n)) for n in ("zero_", "fill_")]
try:
    step(*batch)
    torch.cuda.synchronize()
finally:
    for m, n, o in saved:
        setattr(m, n, o)
for w, c in sites.most_common(40):
    print(f"{c:4d}x {bytes_[w] / 1e6:9.2f} MB  {w}")
