"""Which Python call sites still issue ATen copies / fills / adds during one eager training step?  (monkeypatched counters)"""
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
counts = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/torch/" not in fr.filename and "copy_sites" not in fr.filename and fr.name not in ("nhwc_empty", "_zeroed_f32"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(owner, name, tag, pred=lambda *a, **k: True):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ON[0] and pred(*a, **k):
            counts[(tag, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


cuda = lambda t, *a, **k: isinstance(t, torch.Tensor) and t.is_cuda
wrap(torch.Tensor, "clone", "clone", cuda)
wrap(torch.Tensor, "copy_", "copy_", cuda)
wrap(torch.Tensor, "contiguous", "contiguous(copy)", lambda t, *a, **k: cuda(t) and not t.is_contiguous(*a, **k))
wrap(torch.Tensor, "float", "float(cast)", lambda t, *a, **k: cuda(t) and t.dtype != torch.float32)
wrap(torch.Tensor, "zero_", "zero_", cuda)
wrap(torch.Tensor, "fill_", "fill_", cuda)
wrap(torch, "zeros", "zeros", lambda *a, **k: str(k.get("device", "")).startswith("cuda"))
wrap(torch, "zeros_like", "zeros_like", cuda)
wrap(torch, "empty_like", "empty_like(0)", lambda *a, **k: False)
wrap(torch.Tensor, "to", "to", lambda t, *a, **k: isinstance(t, torch.Tensor) and not t.is_cuda)
ON[0] = True
step(*batch)
torch.cuda.synchronize()
ON[0] = False
for (tag, s), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}x {tag:18s} {s}")
