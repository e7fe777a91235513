"""Runs of consecutive launches of one kernel in a rocprofv3 kernel trace: length, what precedes / follows, grid sizes.
usage: trace_runs.py <kernel_trace.csv> [name]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "")[:50]
i, runs = 0, collections.Counter()
while i < len(rows):
    if pat in rows[i]["Kernel_Name"]:
        j = i
        while j < len(rows) and pat in rows[j]["Kernel_Name"]:
            j += 1
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[j]["Kernel_Name"]) if j < len(rows) else "-"
        grids = collections.Counter(r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in rows[i:j])
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i:j]) / 1e3
        span = (int(rows[j - 1]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        runs[(j - i, prev, nxt, tuple(sorted(grids.items())[:4]), round(dur), round(span))] += 1
        i = j
    else:
        i += 1
for (n, p, q, g, d, s), c in sorted(runs.items(), key=lambda kv: -kv[0][0] * kv[1]):
    print(f"{c:3d} run(s) of {n:4d}: after {p:50s} before {q:50s} busy {d} us span {s} us grids {g}")
