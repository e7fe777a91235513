"""Timeline of the wgrad9x workgroups (DIAGNOSTIC build: octave_amd/csrc/build.sh diag; s_memrealtime stamps at item begin, main
loop begin / end and after the epilogue's atomics have drained): where the time of a launch goes -- prologue, main loop, epilogue,
idle at the end -- per XCD and in total, for the launches of tools/wgrad_sched.py.  Usage (GPU box):
    OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so python tools/wgrad_timeline.py [mode=2] [launch substring ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from octave_amd._lib import lib
from tools.wgrad_sched import LAUNCHES, make, job_array, st

assert "diag" in os.environ.get("OCTA_HIP_LIB", ""), "run with OCTA_HIP_LIB=octave_amd/libocta_hip_diag.so"
dev = torch.device("cuda:0")
L = lib()
dll = L._dll
MAXIT = 12
ROW = 1 + 6 * MAXIT


def read():
    buf = (ctypes.c_uint64 * (2048 * ROW))()
    assert dll.octa_diag_timeline_read_wgrad9x(buf, 0) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(2048, ROW).astype(np.float64)


def main():
    mode = 2
    pick = []
    for a in sys.argv[1:]:
        if a.startswith("mode="):
            mode = int(a[5:])
        else:
            pick.append(a)
    evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    cache = {}
    for lname, lst in LAUNCHES.items():
        if pick and not any(p in lname for p in pick):
            continue
        items = []
        for n in lst:
            if n not in cache:
                cache[n] = make(n)
            items.append(cache[n])
        dws = [torch.zeros_like(it["w"]) for it in items]
        dbs = [torch.zeros(it["w"].shape[0], device=dev) if it["bias"] else None for it in items]
        arr = job_array(items, dws, dbs)
        L.octa_tuning_set(8, mode)
        for _ in range(3):
            L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st())
        torch.cuda.synchronize()
        evict.zero_()
        torch.cuda.synchronize()
        dll.octa_diag_timeline_read_wgrad9x(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.octa_conv2d_wgrad_batch(arr, len(items), None, 0, st())
        e1.record(); e1.synchronize()
        wall = e0.elapsed_time(e1) * 1e3
        a = read()
        n_it = a[:, 0].astype(int)
        wgs = np.nonzero(n_it > 0)[0]
        if len(wgs) == 0:
            print(f"{lname}: no stamps (mode {mode})")
            continue
        R = a[:, 1:].reshape(2048, MAXIT, 6)
        T = R[:, :, :4] * 0.01          # us
        t0 = min(T[w, 0, 0] for w in wgs)
        tend = max(T[w, min(n_it[w], MAXIT) - 1, 3] for w in wgs)
        pro = loop = epi = idle = gap = 0.0
        ends = []
        for w in wgs:
            k = min(n_it[w], MAXIT)
            pro += sum(T[w, i, 1] - T[w, i, 0] for i in range(k))
            loop += sum(T[w, i, 2] - T[w, i, 1] for i in range(k))
            epi += sum(T[w, i, 3] - T[w, i, 2] for i in range(k))
            gap += (T[w, 0, 0] - t0) + sum(T[w, i + 1, 0] - T[w, i, 3] for i in range(k - 1))
            idle += tend - T[w, k - 1, 3]
            ends.append(T[w, k - 1, 3] - t0)
        nw = len(wgs)
        span = tend - t0
        print(f"{lname} | mode {mode} | launch {wall:7.1f} us, first item begin -> last epilogue drained {span:7.1f} us, {nw} workgroups, {int(n_it[wgs].sum())} items "
              f"({n_it[wgs].min()}-{n_it[wgs].max()} per workgroup)")
        print(f"   per-workgroup averages: start skew + gaps {gap / nw:6.1f} us | prologue {pro / nw:6.1f} | main loop {loop / nw:7.1f} | epilogue (atomics drained) {epi / nw:6.1f} | idle at the end {idle / nw:6.1f}"
              f" | = {100 * loop / nw / span:4.1f} % of the span in the main loop")
        ends = np.array(ends)
        print(f"   workgroup end times: min {ends.min():7.1f} p10 {np.percentile(ends, 10):7.1f} median {np.median(ends):7.1f} p90 {np.percentile(ends, 90):7.1f} max {ends.max():7.1f} us")
        for x in range(8):
            ws = [w for w in wgs if w % 8 == x]
            if not ws:
                continue
            row = []
            kmax = max(min(n_it[w], MAXIT) for w in ws)
            for i in range(kmax):
                ww = [w for w in ws if n_it[w] > i]
                b = np.median([T[w, i, 0] - t0 for w in ww]); l = np.median([T[w, i, 2] - T[w, i, 1] for w in ww])
                p = np.median([T[w, i, 1] - T[w, i, 0] for w in ww]); e = np.median([T[w, i, 3] - T[w, i, 2] for w in ww])
                c = np.median([(R[w, i, 5] - R[w, i, 4]) / max(T[w, i, 2] - T[w, i, 1], 1e-3) / 1e3 for w in ww])      # GHz
                row.append(f"@{b:6.0f} p{p:4.1f} L{l:6.1f} {c:4.2f}GHz e{e:5.1f}")
            print(f"   xcd {x}: " + " | ".join(row))
    L.octa_tuning_set(8, 0)


if __name__ == "__main__":
    main()
