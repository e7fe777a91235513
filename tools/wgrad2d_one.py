"""A few launches of one 3x3 layer's weight gradient on wgrad9 (mode 0) or wgrad2d (mode 1), for PMC passes.
Usage: python tools/wgrad2d_one.py <layer> <mode> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd._lib import lib
from tools.wgrad_sched import make, job_array, st

L = lib()
name, mode = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
it = make(name)
dw = torch.zeros_like(it["w"])
arr = job_array([it], [dw], [None])
evict = torch.empty(512 << 20, dtype=torch.uint8, device="cuda:0")
L.octa_tuning_set(10, mode)
for _ in range(n):
    evict.zero_()
    L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
torch.cuda.synchronize()
print(L.octa_last_conv_kernel().decode())
