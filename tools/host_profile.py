"""cProfile of the Python side of eager steps (where do the ~35 ms of host time per step go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid
dev = torch.device("cuda", 0)
B, H = 16, 400
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
x, ys, real = synth_batch(B, H, 0, dev)
pyr = mask_pyramid(real)
for _ in range(3):
    step(x, ys, pyr)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(x, ys, pyr)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
