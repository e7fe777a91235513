"""HBM-traffic model of the conv_halo8 launches of one training step (BASELINE configs[2]: B = 16, 400 x 400, bf16), CPU only.
Per launch: the gathered tensor read once WITH the patch halo (the kernel's own patch choice, clipped to the image), the output
written once, the packed weights once per XCD (8 private L2s), and -- for launches with a tail split -- the raw fp32 partial tiles
the split parts store for halo8_splitk_fix_kernel.  Compares the launch-averaged sum with the PMC measurement
(profiles/r04_pmc_traffic.json) to say how much of the measured traffic is structural and how much is re-reads.
Usage: python tools/halo8_traffic_model.py > profiles/r04_halo8_traffic_model.txt"""
import json
import os

H8_AROWS = 352


def cdiv(a, b):
    return (a + b - 1) // b


def halo8_patch(H, W):                      # halo8.hpp: halo8_patch
    best, PH, PW = -1, 16, 16
    for pw in range(4, 65):
        ph = 256 // pw
        while ph > 1 and (ph + 2) * (pw + 2) > H8_AROWS:
            ph -= 1
        ph = min(ph, H)
        pwe = min(pw, W)
        tiles = cdiv(H, ph) * cdiv(W, pwe)
        score = tiles * 100000 + (ph + 2) * (pwe + 2)
        if best < 0 or score < best:
            best, PH, PW = score, ph, pwe
    return PH, PW


def halo_pixels(H, W, PH, PW):
    n = 0
    for y0 in range(0, H, PH):
        r = min(H, y0 + PH + 1) - max(0, y0 - 1)
        for x0 in range(0, W, PW):
            n += r * (min(W, x0 + PW + 1) - max(0, x0 - 1))
    return n


# (kind, launches per step, H, W, gathered channels Cx, output channels Cy, groups, tail tiles, parts)   -- profiles/r04_layer_times.txt
LAUNCHES = [
    ("dgrad", 1, 100, 100, 256, 512, 1, 0, 1), ("dgrad", 1, 50, 50, 512, 1024, 1, 0, 1), ("fwd", 1, 100, 100, 512, 256, 1, 0, 1),
    ("fwd", 1, 50, 50, 1024, 512, 1, 128, 2), ("fwd", 1, 100, 100, 256, 512, 4, 0, 1), ("fwd", 1, 50, 50, 512, 1024, 4, 0, 1),
    ("dgrad", 1, 50, 50, 1024, 512, 4, 128, 2), ("fwd", 5, 25, 25, 256, 512, 2, 0, 1), ("fwd", 1, 25, 25, 1024, 2048, 4, 0, 1),
    ("fwd", 3, 50, 50, 128, 256, 2, 0, 1), ("fwd", 1, 100, 100, 128, 256, 2, 0, 1), ("dgrad", 2, 13, 13, 1024, 512, 2, 64, 4),
    ("fwd", 2, 13, 13, 512, 1024, 2, 128, 2), ("dgrad", 1, 50, 50, 512, 256, 2, 64, 2), ("fwd", 1, 50, 50, 256, 512, 2, 0, 1),
    ("dgrad", 1, 26, 26, 1024, 512, 2, 0, 1),
]
B = 16
tot_alg = tot_model = n = 0
print(f"{'launch':44s} {'patch':>7s} {'halo':>5s} {'x MB':>7s} {'y MB':>7s} {'w x8 MB':>8s} {'partials MB':>11s} {'model MB':>9s} {'algorithmic MB':>14s} {'ratio':>6s}")
for kind, cnt, H, W, Cx, Cy, g, tail, parts in LAUNCHES:
    PH, PW = halo8_patch(H, W)
    hf = halo_pixels(H, W, PH, PW) / (H * W)
    x, y = B * H * W * Cx * 2, B * H * W * Cy * 2
    w = (Cx // g) * Cy * 9 * 2
    part = tail * parts * 256 * 128 * 4
    model = x * hf + y + 8 * w + part
    alg = x + y + w
    tot_alg += cnt * alg; tot_model += cnt * model; n += cnt
    print(f"{kind:5s} {cnt}x {H:3d}x{W:<3d} {Cx:4d}->{Cy:<4d} g{g} {'+tail%dx%d' % (tail, parts) if tail else '':12s} {PH:3d}x{PW:<3d} {hf:5.2f} {x / 1e6:7.1f} {y / 1e6:7.1f} {8 * w / 1e6:8.1f} {part / 1e6:11.1f} "
          f"{model / 1e6:9.1f} {alg / 1e6:14.1f} {model / alg:6.2f}")
print(f"\nper launch, averaged over the {n} launches of a step: algorithmic {tot_alg / n / 1e6:.1f} MB, model {tot_model / n / 1e6:.1f} MB = {tot_model / tot_alg:.2f} x")
pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_pmc_traffic.json")
if os.path.exists(pmc):
    k = json.load(open(pmc))["kernels"].get("conv_halo8_kernel<bf16,256x128>")
    if k:
        m = k["hbm_bytes_per_launch"]
        print(f"measured (PMC, {os.path.basename(pmc)}): {m / 1e6:.1f} MB per launch = {m / (tot_alg / n):.2f} x algorithmic, {m / (tot_model / n):.2f} x the model")
        print(f"  => of the {m / 1e6 - tot_alg / n / 1e6:.1f} MB above the algorithmic bytes, {tot_model / n / 1e6 - tot_alg / n / 1e6:.1f} MB are structural (halo rows, weights per XCD, "
              f"split-K partial tiles) and {m / 1e6 - tot_model / n / 1e6:.1f} MB are re-reads that missed L2")
