"""Per-kernel totals of two rocprofv3 kernel traces (rocpd .db): python tools/prof_diff.py old.db new.db [steps]"""
import re, sqlite3, sys
a, b = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 14.0


def load(f):
    cur = sqlite3.connect(f).cursor()
    return {re.sub(r"\(.*", "", r[0])[:90]: (r[1], r[2] / 1e6) for r in cur.execute("select name, count(*), sum(end-start) from kernels group by name")}


A, B = load(a), load(b)
print(f"kernel time per step: old {sum(v[1] for v in A.values()) / steps:.2f} ms, new {sum(v[1] for v in B.values()) / steps:.2f} ms  ({steps:.0f} executed steps)")
rows = sorted(((B.get(n, (0, 0))[1] - A.get(n, (0, 0))[1], n) for n in set(A) | set(B)), key=lambda t: -abs(t[0]))
for d, n in rows[:18]:
    ca, ta = A.get(n, (0, 0)); cb, tb = B.get(n, (0, 0))
    print(f"{d / steps * 1e3:+9.1f} us/step  {n[:70]:70s} {ca:5d}->{cb:5d} calls  {ta / steps:7.3f} -> {tb / steps:7.3f} ms/step")
