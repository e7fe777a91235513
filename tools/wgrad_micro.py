"""Correctness + timing of the batched 8-wave weight-gradient kernel (octa_conv2d_wgrad_batch) against the single-problem
kernels (octa_conv2d_wgrad) on real layer shapes, one job per launch and whole stages per launch.
"ab": interleaved A/B of the two batched tile families (octa_tuning_set(1, mask): 1 = 256x128 / 128x256 wgrad8, 3 = 256x256 wgrad9
where it wastes < 15 % padding), 5 rounds each, behind a cache-evicting sweep, median and minimum.
Usage: python tools/wgrad_micro.py [check] [time] [batch] [ab]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib, WgradJob

dev = torch.device("cuda:0")
L = lib()

# name: (B, Cin, H, W, Cout, k, stride, pad, groups, bias)
LAYERS = {
    "dec2_3x3": (16, 512, 100, 100, 256, 3, 1, 1, 1, 0),
    "dec3_3x3": (16, 1024, 50, 50, 512, 3, 1, 1, 1, 0),
    "dec4_3x3": (16, 2048, 25, 25, 1024, 3, 1, 1, 1, 0),
    "dec2_splat": (16, 256, 100, 100, 512, 3, 1, 1, 4, 1),
    "dec3_splat": (16, 512, 50, 50, 1024, 3, 1, 1, 4, 1),
    "dec4_splat": (16, 1024, 25, 25, 2048, 3, 1, 1, 4, 1),
    "dec4_1x1": (16, 2048, 25, 25, 1024, 1, 1, 0, 1, 0),
    "dec2_1x1": (16, 512, 100, 100, 256, 1, 1, 0, 1, 0),
    "enc3_c1": (16, 1024, 25, 25, 256, 1, 1, 0, 1, 0),
    "enc3_splat": (16, 256, 25, 25, 512, 3, 1, 1, 2, 0),
    "enc3_c3": (16, 256, 25, 25, 1024, 1, 1, 0, 1, 0),
    "enc2_c1": (16, 512, 50, 50, 128, 1, 1, 0, 1, 0),
    "enc2_splat": (16, 128, 50, 50, 256, 3, 1, 1, 2, 0),
    "enc2_c3": (16, 128, 50, 50, 512, 1, 1, 0, 1, 0),
    "enc4_c1": (16, 2048, 13, 13, 512, 1, 1, 0, 1, 0),
    "enc4_splat": (16, 512, 13, 13, 1024, 3, 1, 1, 2, 0),
    "enc4_c3": (16, 512, 13, 13, 2048, 1, 1, 0, 1, 0),
    "disc3": (16, 15, 50, 50, 512, 4, 2, 1, 1, 1),
    "up4_adj": (16, 1024, 26, 26, 2048, 2, 2, 0, 1, 0),
    # tall-skinny layers (fewer than 128 output channels per group): single-problem kernels by default
    "disc0": (16, 2, 400, 400, 64, 4, 2, 1, 1, 1),
    "enc1_splat": (16, 64, 100, 100, 128, 3, 1, 1, 2, 0),
    "dec1_splat": (16, 64, 200, 200, 128, 3, 1, 1, 4, 1),
    "dec0_1x1": (16, 64, 400, 400, 32, 1, 1, 0, 1, 0),
    "stem0": (16, 3, 400, 400, 32, 3, 2, 1, 1, 0),
    "stem2": (16, 32, 200, 200, 64, 3, 1, 1, 1, 0),
    "up0_adj": (16, 64, 400, 400, 64, 2, 2, 0, 1, 0),
    "sq1": (16, 64, 200, 200, 13, 1, 1, 0, 1, 1),
}


def make(name, small=None):
    B, Cin, H, W, Cout, k, s, p, g, bias = small or LAYERS[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True)
    x.normal_()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = F_.nhwc_empty(B, Cout, OH, OW, torch.bfloat16, dev, zero=True)
    dy.normal_()
    w = torch.empty(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    d = F_._desc(B, H, W, OH, OW, Cin, Cout, k, k, s, p, g, F_.nhwc_ld(x), F_.nhwc_ld(dy), torch.bfloat16)
    return dict(name=name, x=x, dy=dy, w=w, d=d, bias=bias, cfg=(B, Cin, H, W, Cout, k, s, p, g),
                flops=2.0 * B * OH * OW * Cout * (Cin // g) * k * k)


def job_array(items, dws, dbs):
    arr = (WgradJob * len(items))()
    for j, (it, dw, db) in enumerate(zip(items, dws, dbs)):
        ctypes.memmove(ctypes.byref(arr[j].d), ctypes.byref(it["d"]), ctypes.sizeof(it["d"]))
        arr[j].x, arr[j].dy, arr[j].dw = it["x"].data_ptr(), it["dy"].data_ptr(), dw.data_ptr()
        arr[j].dbias = db.data_ptr() if db is not None else None
        for a in range(4):
            arr[j].dw_strides[a] = dw.stride(a)
    return arr


def st():
    return torch.cuda.current_stream().cuda_stream


def run_old(it, dw, db):
    L.octa_conv2d_wgrad(ctypes.byref(it["d"]), it["x"].data_ptr(), it["dy"].data_ptr(), dw.data_ptr(), F_._strides4(dw),
                        db.data_ptr() if db is not None else None, st())


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def check(it):
    dw_o = torch.zeros_like(it["w"]); dw_n = torch.zeros_like(it["w"])
    Cout = it["cfg"][4]
    db_o = torch.zeros(Cout, device=dev) if it["bias"] else None
    db_n = torch.zeros(Cout, device=dev) if it["bias"] else None
    run_old(it, dw_o, db_o)
    L.octa_conv2d_wgrad_batch(job_array([it], [dw_n], [db_n]), 1, None, 0, st())
    torch.cuda.synchronize()
    scale = dw_o.abs().max().item()
    err = (dw_o - dw_n).abs().max().item()
    msg = f"{it['name']:12s} max|old-new| {err:.3e} of scale {scale:.3e}"
    ok = err <= 2e-4 * scale and scale > 0
    if db_o is not None:
        be = (db_o - db_n).abs().max().item(); bs = db_o.abs().max().item()
        msg += f"  bias {be:.3e} of {bs:.3e}"
        ok = ok and be <= 2e-4 * bs
    print(msg, "OK" if ok else "MISMATCH", flush=True)
    return ok


def check_small():
    """small odd shapes against torch's CPU conv gradients (fp32 math on the bf16-rounded operands)"""
    ok = True
    for name, cfg in {
        "s_1x1": (3, 24, 9, 11, 136, 1, 1, 0, 1, 1),
        "s_3x3": (2, 40, 13, 10, 130, 3, 1, 1, 1, 1),
        "s_3x3s2": (2, 16, 15, 17, 256, 3, 2, 1, 1, 0),
        "s_g2": (2, 32, 12, 12, 256, 3, 1, 1, 2, 1),
        "s_k4s2": (2, 15, 20, 20, 128, 4, 2, 1, 1, 1),
        "s_big_k": (2, 320, 9, 9, 384, 3, 1, 1, 1, 0),
    }.items():
        it = make(name, cfg)
        B, Cin, H, W, Cout, k, s, p, g, bias = cfg
        dw = torch.zeros_like(it["w"])
        db = torch.zeros(Cout, device=dev) if bias else None
        L.octa_conv2d_wgrad_batch(job_array([it], [dw], [db]), 1, None, 0, st())
        xr = it["x"].float().cpu().contiguous(); dyr = it["dy"].float().cpu().contiguous()
        wr = torch.zeros(Cout, Cin // g, k, k, requires_grad=True)
        y = torch.nn.functional.conv2d(xr, wr, None, s, p, 1, g)
        y.backward(dyr)
        scale = wr.grad.abs().max().item()
        err = (dw.cpu() - wr.grad).abs().max().item()
        good = err <= 2e-4 * scale
        msg = f"{name:10s} vs torch CPU: max err {err:.3e} of {scale:.3e}"
        if bias:
            be = (db.cpu() - dyr.sum((0, 2, 3))).abs().max().item()
            msg += f" bias err {be:.3e}"
            good = good and be <= 1e-3 * dyr.sum((0, 2, 3)).abs().max().item() + 1e-3
        print(msg, "OK" if good else "MISMATCH", flush=True)
        ok = ok and good
    return ok


def main():
    args = sys.argv[1:] or ["check", "time", "batch"]
    names = [a for a in args if a in LAYERS] or list(LAYERS)
    ok = True
    if "check" in args:
        ok = check_small() and ok
        for n in names:
            ok = check(make(n)) and ok
    if "time" in args:
        for n in names:
            it = make(n)
            dw = torch.zeros_like(it["w"])
            db = torch.zeros(it["cfg"][4], device=dev) if it["bias"] else None
            arr = job_array([it], [dw], [db])
            t_old = timeit(lambda: run_old(it, dw, db))
            t_new = timeit(lambda: L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st()))
            print(f"{n:12s} old {t_old:8.1f} us {it['flops'] / t_old / 1e6:7.1f} TF/s | new {t_new:8.1f} us {it['flops'] / t_new / 1e6:7.1f} TF/s", flush=True)
    if "batch" in args:
        stages = {
            "enc3 x6": ["enc3_c1", "enc3_splat", "enc3_c3"] * 6,
            "enc2 x4": ["enc2_c1", "enc2_splat", "enc2_c3"] * 4,
            "enc4 x3": ["enc4_c1", "enc4_splat", "enc4_c3"] * 3,
            "dec4": ["dec4_3x3", "dec4_splat", "dec4_1x1", "up4_adj"],
            "dec2+3": ["dec2_3x3", "dec2_splat", "dec2_1x1", "dec3_3x3", "dec3_splat"],
        }
        for sname, lst in stages.items():
            items = [make(n) for n in lst]
            dws = [torch.zeros_like(it["w"]) for it in items]
            dbs = [torch.zeros(it["cfg"][4], device=dev) if it["bias"] else None for it in items]
            arr = job_array(items, dws, dbs)
            fl = sum(it["flops"] for it in items)

            def old():
                for it, dw, db in zip(items, dws, dbs):
                    run_old(it, dw, db)
            t_old = timeit(old)
            t_new = timeit(lambda: L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st()))
            print(f"stage {sname:10s} {len(items):2d} jobs: old {t_old:8.1f} us {fl / t_old / 1e6:7.1f} TF/s | batched {t_new:8.1f} us {fl / t_new / 1e6:7.1f} TF/s", flush=True)
    if "ab" in args:
        evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        sets = {n: [n] for n in names if n in ("dec2_3x3", "dec3_3x3", "dec4_3x3", "dec2_splat", "dec3_splat", "dec4_1x1", "dec2_1x1", "enc3_c3", "enc4_c3", "up4_adj")}
        sets["enc3 x6"] = ["enc3_c1", "enc3_splat", "enc3_c3"] * 6
        sets["enc4 x3"] = ["enc4_c1", "enc4_splat", "enc4_c3"] * 3
        sets["dec4 stage"] = ["dec4_3x3", "dec4_splat", "dec4_1x1", "up4_adj"]
        sets["dec2+3"] = ["dec2_3x3", "dec2_splat", "dec2_1x1", "dec3_3x3", "dec3_splat"]
        for sname, lst in sets.items():
            items = [make(n) for n in lst]
            dws = [torch.zeros_like(it["w"]) for it in items]
            dbs = [torch.zeros(it["cfg"][4], device=dev) if it["bias"] else None for it in items]
            arr = job_array(items, dws, dbs)
            fl = sum(it["flops"] for it in items)
            res = {1: [], 3: []}
            names_k = {}
            for rnd in range(5):
                for mask in (1, 3):
                    L.octa_tuning_set(1, mask)
                    for cold in (True,):
                        evict.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st())
                        e1.record(); e1.synchronize()
                        res[mask].append(e0.elapsed_time(e1) * 1e3)
                        names_k[mask] = L.octa_last_conv_kernel().decode()
            L.octa_tuning_set(1, 3)
            out = []
            for mask in (1, 3):
                v = sorted(res[mask])
                out.append(f"{names_k[mask]:28s} median {v[len(v) // 2]:8.1f} us ({fl / v[len(v) // 2] / 1e6:7.1f} TF/s) min {v[0]:8.1f}")
            print(f"A/B {sname:12s} | " + " | ".join(out), flush=True)
    if "skinny" in args:
        evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        for n in ("disc0", "enc1_splat", "dec1_splat", "dec0_1x1", "stem0", "stem2", "up0_adj", "sq1"):
            it = make(n)
            Cout = it["cfg"][4]
            res = {}
            for mode, minng in (("single-problem kernel", 128), ("batched 8-wave kernel", 8)):
                L.octa_tuning_set(3, minng)
                dw = torch.zeros_like(it["w"])
                db = torch.zeros(Cout, device=dev) if it["bias"] else None
                arr = job_array([it], [dw], [db])
                ts = []
                for _ in range(5):
                    evict.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
                    e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                dw.zero_()
                if db is not None:
                    db.zero_()
                L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
                torch.cuda.synchronize()
                res[mode] = (sorted(ts)[2], dw.clone(), L.octa_last_conv_kernel().decode())
            L.octa_tuning_set(3, 128)
            a, b = res["single-problem kernel"], res["batched 8-wave kernel"]
            err = (a[1] - b[1]).abs().max().item() / max(a[1].abs().max().item(), 1e-30)
            print(f"skinny {n:11s} {a[2]:32s} {a[0]:8.1f} us | {b[2]:30s} {b[0]:8.1f} us | max rel diff {err:.1e}", flush=True)
    if "ablate" in args:
        evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        for n in [x for x in names if x in ("dec2_3x3", "dec3_3x3", "dec4_3x3")] or ["dec2_3x3"]:
            it = make(n)
            dw = torch.zeros_like(it["w"])
            arr = job_array([it], [dw], [None])
            L.octa_tuning_set(1, 3)
            for abl, what in ((0, "full kernel"), (1, "no epilogue atomics"), (2, "no LDS-DMA in the loop"), (4, "no MFMA"), (8, "no transposed reads"),
                              (3, "no atomics, no DMA"), (11, "MFMA only (no atomics / DMA / reads)"), (7, "reads only (no atomics / DMA / MFMA)"),
                              (13, "DMA only (no atomics / MFMA / reads)"), (16, "full, every WG on pixel range 0"),
                              (21, "no atomics / MFMA, pixel range 0"), (29, "DMA only, pixel range 0")):
                L.octa_tuning_set(2, abl)
                ts = []
                for cold in (True, False):
                    for _ in range(4):
                        if cold:
                            evict.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        L.octa_conv2d_wgrad_batch(arr, 1, None, 0, st())
                        e1.record(); e1.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3)
                print(f"ablate {n} {what:40s} cold {sorted(ts[:4])[1]:8.1f} us  warm {sorted(ts[4:])[1]:8.1f} us", flush=True)
            L.octa_tuning_set(2, 0)
    print("ALL OK" if ok else "FAILURES", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
