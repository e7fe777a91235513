"""Which kernels surround the runtime's copyBuffer launches in a rocprofv3 kernel trace?  usage: trace_neighbors.py <kernel_trace.csv> [name]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "")[:60]
ctx = collections.Counter()
sizes = collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        ctx[(prev, nxt)] += 1
        sizes[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))] += 1
print("total", sum(ctx.values()))
for (p, n), c in ctx.most_common(40):
    print(f"{c:5d}x  after {p:60s} before {n}")
print(sizes.most_common(10))
