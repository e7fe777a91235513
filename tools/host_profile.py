"""Where does the HOST time of an eagerly launched training step go?  cProfile over 10 steps (no device sync inside)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid
from octave_amd import functional as F_

B, H = 16, 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, compute_dtype=torch.bfloat16)
x, ys, real = bench.synth_batch(B, H, 0, dev)
batch = (x, ys, mask_pyramid(real))
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    F_.load_algo_cache(sys.argv[1])
step.capture(*batch)
step.launch = "eager"
for _ in range(3):
    step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(*batch)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
