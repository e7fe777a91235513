"""How well-conditioned is the gradient of one full step at BASELINE size?  Runs the HIP fp32 step three times on the same weights
(clean; clean again = run-to-run float-atomic noise only; input image rounded to bf16 = ONE bf16 rounding at the very front) and
the bf16 step, and prints per gradient bucket |g|/|g_clean| and the cosine to the clean fp32 gradient.
Usage: python tools/bf16_conditioning.py [H [adversarial 0/1]]"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from octave_amd import functional as F_                       # noqa: E402
from octave_amd.train import TrainStep, mask_pyramid          # noqa: E402
from test_train_step import _net                              # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    adv = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    dev = torch.device("cuda:0")
    B = 16
    x, ys, real = F_.synth_octa_batch(B, H, H, seed=77, device=dev, vessel=True)
    pyr = mask_pyramid(real)
    runs = [("fp32 clean", torch.float32, x), ("fp32 again", torch.float32, x), ("fp32, x rounded to bf16", torch.float32, x.bfloat16().float()),
            ("bf16", torch.bfloat16, x)]
    ref, bk = None, None
    for tag, dt, xin in runs:
        torch.manual_seed(0)
        net = _net(B, H, dev, seed_fill=False)
        if adv and net.discriminator._has_noise:
            net.discriminator.stack_0[0].is_training = False
        st = TrainStep(net, lr=0.0, compute_dtype=dt, adversarial=adv)
        try:
            torch.manual_seed(3)
            o = st(xin, ys, pyr if adv else None)
            torch.cuda.synchronize()
            g = st.seg_arena.g.double().clone()
            bk = list(st.seg_arena.buckets)
        finally:
            st.close()
        del net, st
        torch.cuda.empty_cache()
        if ref is None:
            ref = g
        rows = []
        for t, lo, hi in bk:
            a, b = ref[lo:hi], g[lo:hi]
            rows.append(f"{t}: {b.norm().item() / a.norm().item():.4f}, {(a @ b).item() / (a.norm().item() * b.norm().item()):.4f}")
        print(f"[{H} {'adv' if adv else 'seg'}] {tag:26s} loss_seg {float(o['loss_seg']):.5f} | ratio, cos per bucket: " + "; ".join(rows), flush=True)


if __name__ == "__main__":
    main()
