"""The four wgrad9 launches of one training step (BASELINE config 1: B = 16, 400 x 400, bf16; compositions read from an
OCTA_WG_LOG=1 run of bench.py), replayed with random operands under the three schedules of the 256 x 256 weight-gradient kernel
(octa_tuning_set(8, mode): 0 = rounds of one split length, 1 = per-class splits + XCD-interleaved sequences, one block per
workgroup, 2 = the same, persistent; mode 3 here = schedule 0 on v_mfma_f32_16x16x32, octa_tuning_set(9, 1); mode 4 = four waves of 128 x 128, octa_tuning_set(9, 2); mode 5 = the bias-free 3x3 layers on the 2-D patch kernel wgrad2d, octa_tuning_set(10, 1)).  Every timing is one launch behind a 512 MB cache-evicting sweep, the modes alternating;
results of modes 1 / 2 are checked against mode 0 first.
Usage: python tools/wgrad_sched.py [check] [time] [rounds=N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octave_amd import functional as F_
from octave_amd._lib import lib, WgradJob

dev = torch.device("cuda:0")
L = lib()

# (B, Cin, H, W, Cout, k, stride, pad, groups, bias)
C = {
    "d2_1x1b": (16, 256, 100, 100, 256, 1, 1, 0, 1, 0), "d2_3x3": (16, 512, 100, 100, 256, 3, 1, 1, 1, 0), "d2_1x1": (16, 512, 100, 100, 256, 1, 1, 0, 1, 0),
    "d3_1x1": (16, 1024, 50, 50, 512, 1, 1, 0, 1, 0), "d3_splat": (16, 512, 50, 50, 1024, 3, 1, 1, 4, 1), "d3_3x3": (16, 1024, 50, 50, 512, 3, 1, 1, 1, 0),
    "d4_1x1": (16, 2048, 25, 25, 1024, 1, 1, 0, 1, 0), "d4_splat": (16, 1024, 25, 25, 2048, 3, 1, 1, 4, 1), "d4_3x3": (16, 2048, 25, 25, 1024, 3, 1, 1, 1, 0),
    "e4a_splat": (16, 512, 26, 26, 1024, 3, 1, 1, 2, 0), "e4a_c1": (16, 1024, 26, 26, 512, 1, 1, 0, 1, 0),
    "e3_c3": (16, 256, 25, 25, 1024, 1, 1, 0, 1, 0), "e3_splat": (16, 256, 25, 25, 512, 3, 1, 1, 2, 0), "e3_c1": (16, 1024, 25, 25, 256, 1, 1, 0, 1, 0),
    "b_4096": (16, 4096, 13, 13, 2048, 1, 1, 0, 1, 0), "e4_c3": (16, 512, 13, 13, 2048, 1, 1, 0, 1, 0), "e4_splat": (16, 512, 13, 13, 1024, 3, 1, 1, 2, 0),
    "e4_c1": (16, 2048, 13, 13, 512, 1, 1, 0, 1, 0), "e4_down": (16, 1024, 13, 13, 2048, 1, 1, 0, 1, 0),
    "e3_down": (16, 512, 25, 25, 1024, 1, 1, 0, 1, 0), "e3a_splat": (16, 256, 50, 50, 512, 3, 1, 1, 2, 0), "e3a_c1": (16, 512, 50, 50, 256, 1, 1, 0, 1, 0),
    "e2_c3x": (16, 256, 50, 50, 512, 1, 1, 0, 1, 0),
}
LAUNCHES = {
    "decoder 4/3/2 (11 jobs)": ["d2_1x1b", "d2_3x3", "d2_1x1", "d3_1x1", "d3_splat", "d3_3x3", "d3_1x1", "d4_1x1", "d4_splat", "d4_3x3", "d4_1x1"],
    "encoder 4 + 3 (20 jobs)": ["e4a_splat", "e4a_c1"] + ["e3_c3", "e3_splat", "e3_c1"] * 3 + ["b_4096", "e4_c3", "e4_splat", "e4_c1", "e4_c3", "e4_splat", "e4_c1", "e4_c3", "e4_down"],
    "encoder 3 (10 jobs)": ["e3_c3", "e3_splat", "e3_c1", "e3_c3", "e3_splat", "e3_c1", "e3_c3", "e3_down", "e3a_splat", "e3a_c1"],
    "encoder 2 tail (1 job)": ["e2_c3x"],
}


def make(name):
    B, Cin, H, W, Cout, k, s, p, g, bias = C[name]
    x = F_.nhwc_empty(B, Cin, H, W, torch.bfloat16, dev, zero=True)
    x.normal_()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = F_.nhwc_empty(B, Cout, OH, OW, torch.bfloat16, dev, zero=True)
    dy.normal_()
    w = torch.empty(Cout, Cin // g, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    d = F_._desc(B, H, W, OH, OW, Cin, Cout, k, k, s, p, g, F_.nhwc_ld(x), F_.nhwc_ld(dy), torch.bfloat16)
    return dict(name=name, x=x, dy=dy, w=w, d=d, bias=bias, flops=2.0 * B * OH * OW * Cout * (Cin // g) * k * k)


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


def main():
    args = sys.argv[1:] or ["check", "time"]
    rounds = 5
    for a in args:
        if a.startswith("rounds="):
            rounds = int(a[7:])
    modes = [int(m) for m in os.environ.get("WG_MODES", "0,1,2").split(",")]
    evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ok = True
    cache = {}
    tot = {m: 0.0 for m in modes}
    for lname, lst in LAUNCHES.items():
        items = []
        for n in lst:                       # identical layers of a launch share their operands (the schedule does not care)
            if n not in cache:
                cache[n] = make(n)
            items.append(cache[n])
        dws = [torch.zeros_like(it["w"]) for it in items]
        dbs = [torch.zeros(it["w"].shape[0], device=dev) if it["bias"] else None for it in items]
        arr = job_array(items, dws, dbs)
        fl = sum(it["flops"] for it in items)
        if "check" in args:
            ref = None
            for m in [0] + [m for m in modes if m]:
                L.octa_tuning_set(8, m if m < 3 else 0); L.octa_tuning_set(9, {3: 1, 4: 2}.get(m, 0)); L.octa_tuning_set(10, 1 if m == 5 else 0)
                for t in dws:
                    t.zero_()
                for t in dbs:
                    if t is not None:
                        t.zero_()
                L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st())
                torch.cuda.synchronize()
                got = [t.clone() for t in dws] + [t.clone() for t in dbs if t is not None]
                if m == 0:
                    ref = got
                    continue
                worst = 0.0
                for a_, b_ in zip(ref, got):
                    worst = max(worst, (a_ - b_).abs().max().item() / max(a_.abs().max().item(), 1e-30))
                good = worst <= 2e-4
                ok = ok and good
                print(f"check {lname:26s} mode {m} vs mode 0: max rel diff {worst:.2e} {'OK' if good else 'MISMATCH'}", flush=True)
        if "time" in args:
            res = {m: [] for m in modes}
            kn = {}
            for rnd in range(rounds):
                for m in modes:
                    L.octa_tuning_set(8, m if m < 3 else 0); L.octa_tuning_set(9, {3: 1, 4: 2}.get(m, 0)); L.octa_tuning_set(10, 1 if m == 5 else 0)
                    evict.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st())
                    e1.record(); e1.synchronize()
                    res[m].append(e0.elapsed_time(e1) * 1e3)
                    kn[m] = L.octa_last_conv_kernel().decode()
            out = []
            for m in modes:
                v = sorted(res[m])
                med = v[len(v) // 2]
                tot[m] += med
                out.append(f"mode {m}: median {med:7.1f} us min {v[0]:7.1f} ({fl / med / 1e6:6.1f} TF/s)")
            print(f"{lname:26s} | " + " | ".join(out), flush=True)
    L.octa_tuning_set(8, 0); L.octa_tuning_set(9, 0); L.octa_tuning_set(10, 2)
    if "time" in args:
        print("sum of medians: " + ", ".join(f"mode {m}: {tot[m]:.1f} us" for m in modes), flush=True)
    print("ALL OK" if ok else "FAILURES", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
