"""One batched weight-gradient launch (wgrad9 / wgrad8) on a layer, a few times -- for rocprofv3 --pmc.  Usage: python tools/wgrad_one.py LAYER [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.wgrad_micro import make, job_array, L, st, dev

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
it = make(name)
dw = torch.zeros_like(it["w"])
arr = job_array([it], [dw], [None])
for _ in range(reps):
    L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
torch.cuda.synchronize()
