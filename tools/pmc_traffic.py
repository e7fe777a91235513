"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs of bench.py, gfx950) into per-kernel HBM bytes
per launch, with the MI355X_MICROARCH.md correction: FETCH_SIZE reports half of a wide coalesced read stream.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import json
import re
import sys


def load(d, counter):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return a


def short(name):
    """rocprof kernel name -> the naming bench.py's roofline leg uses."""
    m8 = re.match(r"void conv_igemm8_kernel<(unsigned short|f16_t), (\d+), (\d+), (\d+)>", name)
    if m8:
        return f"conv_igemm8_kernel<{'bf16' if m8.group(1) == 'unsigned short' else 'f16'},{int(m8.group(2)) * 64}x{int(m8.group(3)) * 64}>"
    mr = re.match(r"void conv_res_kernel<(unsigned short|f16_t), (\d+), (\d+), (\d+), (\d+)>", name)
    if mr:      # <T, TAPS, NCH, TN, MODE> -> the name launch_res() reports (256 pixels x TN*16 channels)
        fam = "conv_res3x3_kernel" if int(mr.group(2)) == 9 else "conv_res1x1_kernel"
        return f"{fam}<{'bf16' if mr.group(1) == 'unsigned short' else 'f16'},256x{int(mr.group(4)) * 16}>"
    mh = re.match(r"void (conv_halo8_kernel|conv_halo16_kernel|conv_halo16p_kernel)<(unsigned short|f16_t), (\d+)>", name)
    if mh:
        return f"{mh.group(1)}<{'bf16' if mh.group(2) == 'unsigned short' else 'f16'},256x128>"
    mp = re.match(r"void pwgemm_kernel<(unsigned short|f16_t), (\d+), (\d+)>", name)
    if mp:
        return f"pwgemm_kernel<{'bf16' if mp.group(1) == 'unsigned short' else 'f16'},{int(mp.group(2)) * 64}x{int(mp.group(3)) * 64}>"
    m9 = re.match(r"void wgrad9_kernel<(\d+), (\d+), (\d+)>", name)
    if m9:
        return f"wgrad9_kernel<{'f16' if int(m9.group(1)) else 'bf16'},256x256>"
    m2 = re.match(r"void wgrad2d_kernel<(\d+)>", name)
    if m2:
        return f"wgrad2d_kernel<{'f16' if int(m2.group(1)) else 'bf16'},256x9x32>"
    mw = re.match(r"void wgrad8_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if mw:
        return f"wgrad8_kernel<{'f16' if int(mw.group(3)) else 'bf16'},{int(mw.group(1)) * 64}x{int(mw.group(2)) * 64}>"
    m = re.match(r"void (conv_igemm_kernel|conv_wgrad_kernel|conv3x3_halo_kernel)<(unsigned short|float), (\d+), (\d+), (\d+), (\d+)", name)
    if not m:
        return name.split("(")[0].replace("void ", "")
    k, t, a, b, c, d = m.group(1), ("bf16" if m.group(2) == "unsigned short" else "f32"), *map(int, m.groups()[2:])
    if k == "conv_wgrad_kernel":
        return f"conv_wgrad_kernel<{t},{a * c * 16}>"
    tile = f"{a * c * 16}x{b * d * 16}"
    return f"{'conv_igemm_kernel' if k == 'conv_igemm_kernel' else 'conv3x3_halo_kernel'}<{t},{tile}>"


if __name__ == "__main__":
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, v in f.items():
        ws = w.get(k, [0.0])
        s = short(k)
        e = out.setdefault(s, {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "wl": 0})
        e["launches"] += len(v)
        e["fetch_kib"] += sum(v)
        e["write_kib"] += sum(ws)
        e["wl"] += len(ws)
    res = {}
    for s, e in out.items():
        fetch = e["fetch_kib"] / e["launches"] * 1024.0
        write = e["write_kib"] / max(e["wl"], 1) * 1024.0
        res[s] = {"launches_profiled": e["launches"], "fetch_size_bytes_per_launch": fetch, "write_size_bytes_per_launch": write,
                  "hbm_bytes_per_launch": 2.0 * fetch + write}
    import datetime
    import os
    import subprocess
    try:
        commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except Exception:
        commit = None
    commit = os.environ.get("OCTA_COMMIT", commit)      # the GPU box has no .git: tools/profile_round.sh passes the commit in
    json.dump({"commit": commit, "date": datetime.date.today().isoformat(),
               "note": "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests as 64 B); averages per launch over "
                       "bench.py --no-graph, B=16, 400x400, bf16; separate --pmc passes",
               "kernels": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print("wrote", sys.argv[3], len(res), "kernels")
