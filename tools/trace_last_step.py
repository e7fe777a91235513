"""The launches of the last complete replayed step of a rocprofv3 kernel trace, in start order: offset from the step's first launch,
duration, gap to the previous launch's end, kernel name.  A step ends with step_end_kernel.
usage: trace_last_step.py <kernel_trace.csv> > step.txt"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "step_end_kernel" in r["Kernel_Name"]]
if len(ends) < 2:
    sys.exit("fewer than two step_end_kernel launches in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")
    name = name.split("(")[0][:90]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {name}  g{r.get('Grid_Size_X', r.get('Grid_Size', '?'))} q{r.get('Queue_Id', '?')}")
    prev_end = max(prev_end, e)
