"""Which gradient goes non-finite in fp16, and at which loss scale?  

For each (B, H) runs the adversarial TrainStep in fp16 with a STATIC loss scale S (no update is applied when a gradient is
non-finite, see TrainStep) and reports, per S: the per-parameter list of non-finite gradients in gradient-completion order
(the first entry is the earliest point of the backward pass where the overflow is visible) and the largest finite |grad| / S.
Usage: python tools/fp16_overflow_probe.py [B H [B H ...]]"""
import sys

import torch

sys.path.insert(0, ".")
from architectures.models.octa import OctaScribbleNet          # noqa: E402
from octave_amd import functional as F_                        # noqa: E402
from octave_amd.train import TrainStep, mask_pyramid           # noqa: E402


def probe(B, H, scales, steps=3, dtype=torch.float16):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
    x, ys, real = F_.synth_octa_batch(B, H, H, seed=77, device=dev, vessel=True)
    pyr = mask_pyramid(real)
    snap = {k: v.clone() for k, v in net.state_dict().items()}
    for S in scales:
        net.load_state_dict(snap)
        st = TrainStep(net, lr=1e-4, compute_dtype=dtype, loss_scale=float(S))
        try:
            for it in range(steps):
                out = st(x, ys, pyr)
                torch.cuda.synchronize()
                rows = []
                for arena, tag in ((st.seg_arena, "seg"), (st.disc_arena, "disc")):
                    bad, mx = [], 0.0
                    for n, p in zip(arena.names, arena.params):
                        g = p.grad
                        fin = torch.isfinite(g)
                        if not bool(fin.all()):
                            bad.append(n)
                        else:
                            mx = max(mx, float(g.abs().max()))
                    rows.append((tag, len(bad), bad[:4], mx / S))
                print(f"B{B} H{H} S=2^{S.bit_length() - 1} step {it}: " + "; ".join(f"{t}: {nb} non-finite {b} max|g|/S {m:.3e}" for t, nb, b, m in rows)
                      + f" loss_seg {float(out['loss_seg']):.4f}", flush=True)
        finally:
            st.close()
            F_._PACK_CACHE.clear()


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]] or [2, 64, 16, 400]
    for B, H in zip(a[0::2], a[1::2]):
        probe(B, H, [1 << k for k in (16, 14, 12, 11, 10, 9, 8, 6)] if H <= 128 else [1 << k for k in (16, 13, 11, 10, 8)])
