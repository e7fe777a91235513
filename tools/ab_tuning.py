"""Same-process A/B of library tuning switches (octa_tuning_set) or environment-read host switches on the replayed B = 16, 400 x 400
adversarial step: per round and configuration a fresh TrainStep is captured (the switches are read when a launch is RECORDED) and 30
replays are timed; configurations alternate, the kernel choices are tuned once and shared.
usage (GPU box): python tools/ab_tuning.py "7=0" "7=1" ["7=1,6=0" ...] [rounds]        (KEY=VALUE[,KEY=VALUE...] per configuration; "" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from architectures.models.octa import OctaScribbleNet
from octave_amd import functional as F_
from octave_amd._lib import lib
from octave_amd.train import TrainStep, mask_pyramid

DEFAULTS = {1: 3, 4: 0, 5: 0, 6: 1, 7: 0, 8: 0, 9: 0, 10: 2}
cfgs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
dev = torch.device("cuda", 0)
B, H = 16, 400
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
x, ys, real = synth_batch(B, H, 0, dev)
pyr = mask_pyramid(real)


def apply(cfg):
    for k, v in DEFAULTS.items():
        lib().octa_tuning_set(k, v)
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        lib().octa_tuning_set(int(k), int(v))


res = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        apply(c)
        step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16)
        step.capture(x, ys, pyr)
        for _ in range(5):
            step(x, ys, pyr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step(x, ys, pyr)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        res[c].append(ms)
        print(f"round {r} [{c or 'defaults'}] {ms:.3f} ms/step", flush=True)
        step.close()
        del step
        torch.cuda.empty_cache()
apply("")
for c in cfgs:
    v = sorted(res[c])
    print(f"[{c or 'defaults'}] median {v[len(v) // 2]:.3f} min {v[0]:.3f} ms/step over {len(v)} rounds")
