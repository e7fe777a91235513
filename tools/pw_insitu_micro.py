"""Mid-size pointwise layers: kernel time per algo WARM (the same launch repeated: operands in L2 / the memory-side cache) against IN SITU
(a producer kernel rewrites the input right before every launch, as the BatchNorm apply does in the step, and a second tensor of the
output's size is rewritten too so that the output lines are not resident either).  20 iterations per hipGraph; in situ = chain - producer.
The autotuner (functional._choose_algo) times warm launches; this says whether its ranking holds behind a producer.
Usage: python tools/pw_insitu_micro.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from tools.conv8_micro import LAYERS
from tools.pw_micro import graph_time

dev = torch.device("cuda:0")
ALGOS = tuple(int(a) for a in os.environ.get("PW_ALGOS", "1,4,5,6,2,3,8,9,10,11").split(","))


def run(name):
    B, Cin, H, W, Cout, k, s, p, g = LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True); x.normal_()
    src = torch.randn_like(x)
    w = torch.nn.Parameter((torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    F_._ALGO_OVERRIDE = 1
    y = F_.raw_conv_fwd(x, w, None, s, p, g, 0)
    dy = torch.randn_like(y)
    dsrc = torch.randn_like(dy)
    line = f"{name:11s}"
    for kind in ("fwd", "dgrad"):
        inp, isrc = (x, src) if kind == "fwd" else (dy, dsrc)
        t_prod = graph_time(lambda: inp.copy_(isrc))
        line += f" | {kind} (producer {t_prod:4.1f}):"
        for algo in ALGOS:
            F_._ALGO_OVERRIDE = algo
            conv = (lambda: F_.raw_conv_fwd(x, w, None, s, p, g, 0)) if kind == "fwd" else (lambda: F_.raw_conv_dgrad(dy, w, tuple(x.shape), s, p, g))
            t_warm = graph_time(conv)

            def chain():
                inp.copy_(isrc)
                conv()
            t_chain = graph_time(chain)
            line += f" a{algo} {t_warm:5.1f}/{t_chain - t_prod:5.1f}"
    F_._ALGO_OVERRIDE = 0
    print(line, flush=True)


if __name__ == "__main__":
    print("# layer | kind: per algo warm / in situ (us)")
    names = [a for a in sys.argv[1:] if a in LAYERS] or ["enc2_c1", "enc2_c3", "enc3_c1", "enc3_c3", "enc3_ds", "enc4_c1", "enc4_c3"]
    for n in names:
        run(n)
