"""Fold a rocprofv3 kernel_stats.csv into per-family time per step.  usage: stats_summary.py <csv> <steps executed>"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = collections.OrderedDict([
    ("conv fwd/dgrad", r"conv_igemm|conv3x3_halo_kernel|conv_halo8|conv_halo16|halo8_splitk|igemm8_splitk|pwgemm|conv_res_kernel|col2im"), ("conv wgrad", r"wgrad"), ("split-attention (+ its bn0 on the fly)", r"splat"), ("batchnorm", r"bn_"),
    ("attention gate", r"aag"), ("loss", r"wpce|kl_|softmax|lsgan"), ("pack/adam", r"pack|adam"), ("disc misc", r"noise|spectral|fullconv"),
    ("layout/pool/copy (octa)", r"nchw|nhwc|copy_channels|pool|act_bwd|colsum|zero_words"), ("ATen / runtime", r"at::|rocclr|Cijk|elementwise|vectorized"),
])
agg = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for r in rows:
    name, t, n = r["Name"], float(r["TotalDurationNs"]), int(r["Calls"])
    total += t
    for f, pat in fam.items():
        if re.search(pat, name):
            agg[f][0] += t; agg[f][1] += n
            break
    else:
        agg["other"][0] += t; agg["other"][1] += n
print(f"total kernel time {total / steps / 1e6:.2f} ms/step over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for f, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {f:26s} {t / steps / 1e6:7.2f} ms/step {n / steps:7.0f} launches/step")
print("top kernels:")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f"  {float(r['TotalDurationNs']) / steps / 1e6:6.2f} ms {int(r['Calls']) / steps:6.0f}x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
