"""One layer, one algo, a few launches -- for rocprofv3 --pmc runs.  Usage: python tools/conv_one.py LAYER ALGO [fwd|dgrad] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from tools.conv8_micro import LAYERS

dev = torch.device("cuda:0")
name, algo = sys.argv[1], int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "fwd"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
F_.set_splitk_workspace(ws)
F_._ALGO_OVERRIDE = 1
y = F_.raw_conv_fwd(x, w, None, s, p, g, 0)
dy = torch.randn_like(y)
F_._ALGO_OVERRIDE = algo
for _ in range(reps):
    if kind == "fwd":
        F_.raw_conv_fwd(x, w, None, s, p, g, 0)
    else:
        F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g)
torch.cuda.synchronize()
