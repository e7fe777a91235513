"""Dual-head U-Net backward: HIP fp32 gradients against the CPU oracle in float64 and float32, per top-level module
(relative L2 error of the gradient VECTORS, not only their norms), several HIP runs.  Separates a structural error (a module
whose error is O(1) in every run) from amplified rounding noise (errors of a few per cent that move from run to run and grow
towards the encoder).  Usage: python tests/diag/heads_bwd_diag.py [ph|phag] [runs]"""
import collections
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import ref_ops as R                                # noqa: E402
from oracle.fill import fill_state_dict, hash_input          # noqa: E402
from test_round3 import _cotangent_loss, _head_outputs, _heads  # noqa: E402


def oracle_grads(tag, dt):
    m = _heads(tag)
    fill_state_dict(m.state_dict())
    P = {}
    for k, v in m.state_dict().items():
        v = v.clone().to(dt) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
        P[k] = v
    x = hash_input((3, 1, 48, 48), 1234).repeat(1, 3, 1, 1).to(dt)
    att, att_c, agg = R.parallel_head_forward(x, P, gates=tag == "phag", gating_level=3)
    outs = [agg] if tag == "ph" else [agg, *att, *att_c]
    loss = 0
    for i, o in enumerate(outs):
        loss = loss + (o * hash_input(tuple(o.shape), 7100 + i, -1.0, 1.0).to(dt)).sum()
    loss.backward()
    return {k: v.grad.double() for k, v in P.items() if v.requires_grad and v.grad is not None}


def hip_grads(tag, dev):
    m = _heads(tag)
    fill_state_dict(m.state_dict())
    m = m.to(dev).train()
    x = hash_input((3, 1, 48, 48), 1234).repeat(1, 3, 1, 1).to(dev)
    loss = _cotangent_loss(_head_outputs(tag, m(x)), dev)
    loss.backward()
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "ph"
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    g64, g32 = oracle_grads(tag, torch.float64), oracle_grads(tag, torch.float32)
    hips = [hip_grads(tag, dev) for _ in range(runs)]
    top = max(v.norm().item() for v in g64.values())

    def table(g):
        per = collections.defaultdict(list)
        for k, w in g64.items():
            if w.norm().item() <= 1e-6 * top or k not in g:
                continue
            per[k.split(".", 1)[0]].append((g[k] - w).norm().item() / w.norm().item())
        return {m: (float(np.median(v)), float(np.max(v))) for m, v in per.items()}
    def signed(g):
        per = collections.defaultdict(list)
        for k, w in g64.items():
            if w.norm().item() <= 1e-6 * top or k not in g:
                continue
            per[k.split(".", 1)[0]].append(g[k].norm().item() / w.norm().item() - 1.0)
        return {m: float(np.median(v)) for m, v in per.items()}
    s32, shs = signed(g32), [signed(h) for h in hips]
    print(f"== {tag}: SIGNED gradient-norm deviation |g| / |g64| - 1, median per module")
    for m in s32:
        print(f"{m:16s} oracle fp32 {s32[m]:+9.2e}   HIP " + " ".join(f"{t[m]:+9.2e}" for t in shs))
    t32 = table(g32)
    ths = [table(h) for h in hips]
    print(f"== {tag}: relative L2 error of the gradient per module (median, max over its parameters) vs oracle float64")
    print(f"{'module':16s} {'oracle fp32':>20s} " + " ".join(f"{'HIP run ' + str(i):>20s}" for i in range(runs)))
    for m in t32:
        print(f"{m:16s} {t32[m][0]:9.2e} {t32[m][1]:9.2e}  " + " ".join(f"{t[m][0]:9.2e} {t[m][1]:9.2e} " for t in ths))


if __name__ == "__main__":
    main()
