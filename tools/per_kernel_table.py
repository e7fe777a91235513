"""The `roofline.per_kernel` object of a bench.py JSON line as a markdown table.  usage: per_kernel_table.py <bench.json> > table.md"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"Per-kernel roofline of the conv engine (`{sys.argv[1]}`: {d['ms_per_step']} ms per step; `bench.py`'s roofline leg: every conv-engine launch of one step")
print("replayed once, cold -- behind a 512 MB sweep -- between HIP events, scratch passed as in the step; FLOPs = 2 M N K, bytes = every operand once; peaks")
print("2.5 PFLOP/s dense bf16 and 8 TB/s).  Batched weight-gradient rows include their fold launch.")
print()
print("| kernel instance | launches / step | avg us | ms / step | TFLOP/s | GB/s (algorithmic) | bound | fraction of its roofline |")
print("|---|---|---|---|---|---|---|---|")
pk = r["per_kernel"]
for name in sorted(pk, key=lambda n: -pk[n]["launches_per_step"] * pk[n]["avg_us"]):
    k = pk[name]
    print(f"| `{name}` | {k['launches_per_step']} | {k['avg_us']:.1f} | {k['launches_per_step'] * k['avg_us'] / 1e3:.2f} | {k['tflops']:.0f} | {k['gbs']:.0f} | {k['bound']} | {k['frac']:.3f} |")
a = r.get("all_conv_kernels")
if a:
    print(f"\nAll conv-engine launches: {a['gflop_per_step']:.0f} GFLOP in {a['time_ms_per_step']:.2f} ms (cold) = {a['achieved']:.0f} TFLOP/s = {a['frac']:.3f} of the dense peak.")
