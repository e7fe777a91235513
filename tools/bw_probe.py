import torch
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (164, 328, 656):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    t_copy = timeit(lambda: y.copy_(x)); t_fill = timeit(lambda: y.zero_()); t_read = timeit(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32))
    t_relu = timeit(lambda: torch.relu_(y))
    print(f"{mb} MB: copy {t_copy:.1f}us ({2*mb/t_copy*1e-3:.2f} TB/s)  fill {t_fill:.1f}us ({mb/t_fill*1e-3:.2f} TB/s)  read-sum {t_read:.1f}us ({mb/t_read*1e-3:.2f} TB/s) relu_ {t_relu:.1f}us ({2*mb/t_relu*1e-3:.2f} TB/s)")
