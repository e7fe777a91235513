"""Is the fp16 deviation of the LS-GAN term (tests/test_round4.py::test_adversarial_step_fp16_vs_reference_48) the discriminator's
own fp16 arithmetic or the chaotic segmentor in front of it?  Same attention maps (from the fp32 segmentor) through the
discriminator in fp32, bf16 and fp16; then the segmentor in fp16 / bf16 / fp32-with-rounded-input and the discriminator in fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from architectures.models.octa import OctaScribbleNet
from octave_amd.synth import fill_state_dict, hash_input

dev = torch.device("cuda:0")
Bn, H = 6, 48
x = hash_input((Bn, 1, H, H), 1234).repeat(1, 3, 1, 1).to(dev)


def build(seg_dt, disc_dt):
    net = OctaScribbleNet(torch.Size((Bn, 3, H, H)), torch.Size((Bn, 2, H, H)), True, False, instance_noise=False, label_noise=False)
    fill_state_dict(net.state_dict())
    net = net.to(dev).train()
    net.segmentor.compute_dtype = seg_dt
    net.discriminator.compute_dtype = disc_dt
    return net


with torch.no_grad():
    att32 = build(torch.float32, torch.float32).segmentor(x)[0]
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        f = build(torch.float32, dt).discriminator([a.clone() for a in att32])
        print(f"D in {str(dt):16s} on the fp32 attention maps: f = {[round(v, 4) for v in f.flatten().tolist()]}  g_adv = {0.5 * ((f - 1) ** 2).mean().item():.5f}")
    for tag, dt, xin in (("fp16", torch.float16, x), ("bf16", torch.bfloat16, x), ("fp32, x->fp16 once", torch.float32, x.half().float()), ("fp32, x->bf16 once", torch.float32, x.bfloat16().float())):
        net = build(dt, torch.float32)
        att = net.segmentor(xin)[0]
        f = net.discriminator(att)
        d = max((a.float() - b).abs().max().item() for a, b in zip(att, att32))
        print(f"segmentor {tag:20s} -> D in fp32: max |att - att32| {d:.3e}  f = {[round(v, 4) for v in f.flatten().tolist()]}  g_adv = {0.5 * ((f - 1) ** 2).mean().item():.5f}")
