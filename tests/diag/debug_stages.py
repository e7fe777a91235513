"""Per-stage parity of the HIP segmentor against the oracle: every stage of the oracle is fed the
HIP path's own input to that stage, so errors do not accumulate through the 93 BatchNorms.
Usage (GPU box): python tests/diag/debug_stages.py [H] [B] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_ops as R
from oracle.fill import fill_state_dict, hash_input
from architectures.models.octa import OctaScribbleNet

H = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
dev = torch.device("cuda:0")
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False)
fill_state_dict(net.state_dict())
P = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to(dev).train()
seg = net.segmentor
seg.compute_dtype = dt
cap = {}

def hook(name):
    def f(mod, inp, out):
        cap[name] = (tuple(i.detach().float().cpu() if torch.is_tensor(i) else i for i in inp),
                     tuple(o.detach().float().cpu() for o in (out if isinstance(out, tuple) else (out,))))
    return f
names = ["encoder_0_2_2", "encoder_1", "encoder_2", "encoder_3", "encoder_4"] + [f"{p}_{i}" for i in range(5) for p in ("upsampling", "decoder", "aag")]
for n in names:
    getattr(seg, n).register_forward_hook(hook(n))
for i in range(3):
    seg.encoder_1[i].register_forward_hook(hook(f"encoder_1.{i}"))
seg.encoder_1[0].conv2.register_forward_hook(hook("encoder_1.0.conv2"))
seg.encoder_2[0].register_forward_hook(hook("encoder_2.0"))
seg.decoder_4.conv[3].register_forward_hook(hook("decoder_4.conv.3"))
x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1)
att, agg, x4 = seg(x.to(dev))
torch.cuda.synchronize()

def rep(name, got, want):
    e = (got - want).abs().max().item()
    print(f"{name:22s} max|err| {e:.3e}  max|want| {want.abs().max().item():.3e}  rel {e / (want.abs().max().item() + 1e-30):.2e}")

Q = lambda: {k: v.clone() for k, v in P.items()}
with torch.no_grad():
    rep("stem", cap["encoder_0_2_2"][0][0], R.stem(x, Q(), "segmentor.encoder_0_1_2"))
    rep("maxpool", cap["encoder_0_2_2"][1][0], torch.nn.functional.max_pool2d(cap["encoder_0_2_2"][0][0], 3, 2, 1))
    Q64 = lambda: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
    for i in range(4):
        n = f"encoder_{i+1}"
        w32 = R.encoder_stage(cap[n][0][0], Q(), "segmentor." + n, i)
        w64 = R.encoder_stage(cap[n][0][0].double(), Q64(), "segmentor." + n, i)
        rep(n, cap[n][1][0], w32)
        rep(n + " [cpu32-cpu64]", w32.double(), w64)
        rep(n + " [hip-cpu64]", cap[n][1][0].double(), w64)
    for i in range(3):
        n = f"encoder_1.{i}"
        rep(n, cap[n][1][0], R.bottleneck(cap[n][0][0], Q(), "segmentor." + n, 1, i == 0))
    n = "encoder_1.0.conv2"
    rep(n, cap[n][1][0], R.splat_conv2d(cap[n][0][0], Q(), "segmentor." + n, 1))
    n = "encoder_2.0"
    rep(n, cap[n][1][0], R.bottleneck(cap[n][0][0], Q(), "segmentor." + n, 2, True))
    n = "decoder_4.conv.3"
    rep(n, cap[n][1][0], torch.relu(R.splat_conv2d(cap[n][0][0], Q(), "segmentor." + n, 2)))
    for i in range(5):
        n = f"upsampling_{i}"
        rep(n, cap[n][1][0], R.upsampling(cap[n][0][0], Q(), "segmentor." + n))
        n = f"decoder_{i}"
        rep(n, cap[n][1][0], R.resnest_decoder(cap[n][0][0], Q(), "segmentor." + n))
        n = f"aag_{i}"
        m, y = R.attention_gate(cap[n][0][0], Q(), "segmentor." + n)
        rep(n + " masked", cap[n][1][0], m)
        rep(n + " y", cap[n][1][1], y)
    full = R.resnest_unet_forward(x, Q())[1]
    rep("end-to-end logits", agg.detach().float().cpu(), full)
    P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
    full64 = R.resnest_unet_forward(x.double(), P64)[1]
    rep("cpu fp32 vs fp64", full.double(), full64)
    rep("hip vs fp64", agg.detach().double().cpu(), full64)
