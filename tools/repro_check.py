"""Run-to-run reproducibility of a stage at BASELINE geometry (float atomics are the only legitimate source of differences).
For each module of the stage: forward twice on the same input, backward twice on the same cotangent; report max |diff| / max |value|
of the output, the input gradient and the parameter gradients.  Usage: python tools/repro_check.py [stage ...]"""
import sys

import torch

sys.path.insert(0, ".")
from architectures.segmentor.compose import ResnestUNet       # noqa: E402
from octave_amd import functional as F_                       # noqa: E402

STAGES = {"encoder_3": (16, 512, 50, 50), "decoder_3": (16, 1024, 50, 50), "decoder_2": (16, 512, 100, 100), "encoder_4": (16, 1024, 26, 26),
          "encoder_2": (16, 256, 100, 100)}


def run(mod, x, g=None):
    xin = x.detach().clone().requires_grad_(True)
    for p in mod.parameters():
        p.grad = None
    y = mod(xin)
    y = y[0] if isinstance(y, tuple) else y
    if g is None:
        g = torch.randn(tuple(y.shape), generator=torch.Generator(device="cpu").manual_seed(9)).to(x.device).to(x.dtype)
        g = F_.to_nhwc(g)
    y.backward(g)
    torch.cuda.synchronize()
    return F_.to_nchw_f32(y.detach()), F_.to_nchw_f32(xin.grad), {n: p.grad.detach().float().clone() for n, p in mod.named_parameters() if p.grad is not None}, g


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    unet = ResnestUNet(2, False).to(dev).train()
    for name in (sys.argv[1:] or ["encoder_3", "decoder_3"]):
        for dt in (torch.float32, torch.bfloat16):
            stage = getattr(unet, name)
            x = F_.to_nhwc(torch.randn(*STAGES[name], generator=torch.Generator(device="cpu").manual_seed(5)).to(dev).to(dt))
            subs = list(stage.children()) if isinstance(stage, torch.nn.Sequential) else [stage]
            cur = x
            print(f"== {name} {dt}: {len(subs)} block(s)")
            for bi, m in enumerate([stage] + (subs if len(subs) > 1 else [])):
                sd = {k: v.clone() for k, v in m.state_dict().items()}
                inp = x if m is stage else cur
                y1, dx1, g1, g = run(m, inp)
                m.load_state_dict(sd)
                y2, dx2, g2, _ = run(m, inp, g)
                m.load_state_dict(sd)
                worst = max(((rel(g1[k], g2[k]), k) for k in g1), default=(0.0, ""))
                tag = "whole stage" if m is stage else f"block {bi - 1}"
                print(f"   {tag:12s} y {rel(y1, y2):.2e}  dx {rel(dx1, dx2):.2e}  worst param grad {worst[0]:.2e} ({worst[1]})", flush=True)
                if m is not stage:
                    with torch.no_grad():
                        cur = m(inp).detach()
                    m.load_state_dict(sd)


if __name__ == "__main__":
    main()
