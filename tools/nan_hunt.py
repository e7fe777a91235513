"""Run the bench workload step by step and report the first non-finite loss / parameter / gradient
(--async: enqueue the steps without host synchronisation and log device-side health flags per step).
usage: python tools/nan_hunt.py [--steps N] [--no-graph] [--batch B] [--size S]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from architectures.models.octa import OctaScribbleNet
from octave_amd.train import TrainStep, mask_pyramid

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=400)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--seg-only", action="store_true")
ap.add_argument("--side-stream", action="store_true")
ap.add_argument("--runahead", type=int, default=-1, help="async mode: wait for step i-K before enqueuing step i")
ap.add_argument("--no-clone", action="store_true")
ap.add_argument("--async", dest="asyn", action="store_true", help="enqueue all steps without synchronising (as bench.py does)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, H = args.batch, args.size
torch.manual_seed(0)
net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
step = TrainStep(net, lr=1e-4, compute_dtype=torch.bfloat16, adversarial=not args.seg_only)
x, ys, real = synth_batch(B, H, 0, dev)
batch = (x, ys, mask_pyramid(real))
if not args.no_graph:
    step.capture(*batch)


def report(tag):
    for nm, ar in (("seg", step.seg_arena), ("disc", step.disc_arena))[:1 if args.seg_only else 2]:
        for k in ("p", "g", "m", "v"):
            t = getattr(ar, k)
            if t is not None and not torch.isfinite(t).all():
                print(f"  {tag}: {nm}.{k} has {int((~torch.isfinite(t)).sum())} non-finite of {t.numel()}")
    for mod, nm in ((net.segmentor, "seg"), (net.discriminator, "disc"))[:1 if args.seg_only else 2]:
        bad = [(n, int((~torch.isfinite(p)).sum())) for n, p in mod.named_parameters() if not torch.isfinite(p).all()]
        badg = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in mod.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        badb = [(n, int((~torch.isfinite(b)).sum())) for n, b in mod.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
        print(f"  {tag}: {nm} params with NaN: {len(bad)} {bad[:6]}\n  {tag}: {nm} grads with NaN: {len(badg)} {badg[:6]}\n  {tag}: {nm} buffers with NaN: {len(badb)} {badb[:6]}")


if args.asyn:
    for i in range(5):
        step(*batch)
        torch.cuda.synchronize()
    hist, evs = [], []
    ctx = torch.cuda.stream(torch.cuda.Stream()) if args.side_stream else torch.no_grad()
    if args.side_stream:
        torch.cuda.synchronize()
    with ctx:
        for i in range(args.steps):
            if args.runahead >= 0 and i - args.runahead - 1 >= 0:
                evs[i - args.runahead - 1].synchronize()
            out = step(*batch)
            if not args.no_clone or i == args.steps - 1:
                h = {k: v.clone() for k, v in out.items()}
                fin = lambda t: torch.isfinite(t).all().float()
                h["F_segp"], h["F_segg"], h["F_segm"], h["F_segv"] = fin(step.seg_arena.p), fin(step.seg_arena.g), fin(step.seg_arena.m), fin(step.seg_arena.v)
                if step.disc_arena is not None:
                    h["F_dp"], h["F_dg"], h["F_dm"], h["F_dv"] = fin(step.disc_arena.p), fin(step.disc_arena.g), fin(step.disc_arena.m), fin(step.disc_arena.v)
                    h["vmin_d"] = step.disc_arena.v.min()
                    h["noise_absmax"] = step._caps[H].feed.noise_dev.abs().max()
                    h["sign0"], h["sign1"], h["sign2"] = step._caps[H].feed.sign_dev[0].clone(), step._caps[H].feed.sign_dev[1].clone(), step._caps[H].feed.sign_dev[2].clone()
                    h["dynD0"], h["dynD1"] = step._dyn_dev[1][0].clone(), step._dyn_dev[1][1].clone()
                    for j, a in enumerate(step._att):
                        h[f"F_att{j}"] = fin(a)
                    sd = dict(net.discriminator.named_buffers())
                    for k2, b2 in sd.items():
                        if "weight_u" in k2:
                            h["F_" + k2.split(".")[1] + "_u"] = fin(b2)
                for nm_, ar_ in (("s", step.seg_arena), ("d", step.disc_arena)):
                    if ar_ is None:
                        continue
                    for kk in ("p", "g"):
                        badm = ~torch.isfinite(getattr(ar_, kk))
                        h[f"N_{nm_}{kk}"] = badm.sum().float()
                        h[f"I_{nm_}{kk}"] = badm.float().argmax().float()
                        h[f"L_{nm_}{kk}"] = (badm.numel() - 1 - badm.flip(0).float().argmax()).float()
                h["dynS0"], h["dynS1"] = step._dyn_dev[0][0].clone(), step._dyn_dev[0][1].clone()
                h["vmin_s"] = step.seg_arena.v.min()
                hist.append(h)
            e = torch.cuda.Event(); e.record(); evs.append(e)
    torch.cuda.synchronize()
    bad = False
    for i, h in enumerate(hist):
        vals = {k: float(v) for k, v in h.items()}
        print(i, " ".join(f"{k}={v:.5g}" for k, v in vals.items()), flush=True)
        bad = bad or not all(v == v and abs(v) < 1e30 for v in vals.values())
    if bad:
        report("async end")
        for nm_, ar_ in (("seg", step.seg_arena), ("disc", step.disc_arena)):
            if ar_ is None:
                continue
            base = ar_.p.data_ptr()
            names = {id(p_): n_ for n_, p_ in net.named_parameters()}
            print(nm_, "arena numel", ar_.numel, "layout:", [(names.get(id(p_), "?"), (p_.data_ptr() - base) // 4, p_.numel()) for p_ in ar_.params][:400])
    sys.exit(0)

for i in range(args.steps):
    out = step(*batch)
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in out.items()}
    print(i, " ".join(f"{k}={v:.5f}" for k, v in vals.items()), flush=True)
    if not all(v == v and abs(v) < 1e30 for v in vals.values()):
        report(f"step {i}")
        break
else:
    print("no non-finite value in", args.steps, "steps")
