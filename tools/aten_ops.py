"""Which ATen / runtime ops does one eager training step still launch?  (torch.profiler, with Python stacks for the top offenders)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid
from torch.profiler import profile, ProfilerActivity

B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, compute_dtype=torch.bfloat16)
x, ys, real = bench.synth_batch(B, H, 0, dev)
batch = (x, ys, mask_pyramid(real))
step(*batch); step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(*batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
agg = {}
for e in rows:
    a = agg.setdefault(e.key, [0, 0.0]); a[0] += e.count; a[1] += e.device_time_total
print("== aten ops with device time, per step")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:5d}x {t:9.1f} us  {k}")
print("== by call site")
for e in rows[:60]:
    st = [s for s in e.stack if "site-packages/torch" not in s and "<built-in" not in s][:3]
    print(f"{e.count:4d}x {e.device_time_total:8.1f}us {e.key:28s} | " + " <- ".join(s.split('/')[-1] for s in st))
