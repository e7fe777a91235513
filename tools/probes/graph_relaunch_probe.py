"""ROCm hipGraph behaviour probe: host time of back-to-back replays of ONE graph vs two alternating copies (no sync in between)."""
import time, torch
dev = torch.device("cuda:0")
x = torch.randn(64 << 20, device=dev)          # 256 MB: each kernel ~0.1 ms


def make():
    g = torch.cuda.CUDAGraph()
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            torch.mul(x, 1.0001, out=y)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(300):
            torch.mul(x, 1.0001, out=y)
    return g, y


def run(graphs, n=20, label=""):
    torch.cuda.synchronize()
    ts, evs = [], [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter(); evs[0].record()
    for k in range(n):
        a = time.perf_counter(); graphs[k % len(graphs)].replay(); evs[k + 1].record(); ts.append((time.perf_counter() - a) * 1e3)
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    print(f"{label}: total {tot:.1f} ms/replay host {host:.1f}; host ms: " + " ".join(f"{t:.0f}" for t in ts))
    print("      device ms: " + " ".join(f"{evs[k].elapsed_time(evs[k + 1]):.0f}" for k in range(n)))


g1, _ = make(); g2, _ = make(); g3, _ = make()
for rep in range(2):
    run([g1], label="one graph     ")
    run([g1, g2], label="two alternating")
    run([g1, g2, g3], label="three alternating")

# when does the host SEE an event recorded between queued replays?
print("event visibility: replays queued back to back, events in between, polled with query()")
for nq in (2, 4, 8):
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for k in range(nq):
        g1.replay() if k % 2 == 0 else g2.replay()
        e = torch.cuda.Event(); e.record(); evs.append(e)
    seen = []
    for e in evs:
        while not e.query():
            time.sleep(0.0002)
        seen.append((time.perf_counter() - t0) * 1e3)
    print(f"  {nq} queued: events seen at ms " + " ".join(f"{t:.0f}" for t in seen))
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for k in range(nq):
        g1.replay() if k % 2 == 0 else g2.replay()
        e = torch.cuda.Event(); e.record(); evs.append(e)
    seen = []
    for e in evs:
        e.synchronize()
        seen.append((time.perf_counter() - t0) * 1e3)
    print(f"  {nq} queued: synchronize() returned at ms " + " ".join(f"{t:.0f}" for t in seen))

print("host cost of a pinned H2D hipMemcpyAsync on an IDLE side stream while 3 graph replays are queued on the main stream")
side = torch.cuda.Stream()
for nbytes in (8, 4096, 1 << 16, 1 << 21):
    h = torch.zeros(nbytes // 4, dtype=torch.float32).pin_memory()
    d = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for k in range(3):
        g1.replay() if k % 2 == 0 else g2.replay()
    ts = []
    for rep in range(3):
        a = time.perf_counter()
        with torch.cuda.stream(side):
            d.copy_(h, non_blocking=True)
        ts.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    print(f"  {nbytes:8d} B: host ms per copy " + " ".join(f"{t:.2f}" for t in ts))
print("same, copy issued on the MAIN stream")
for nbytes in (8, 1 << 21):
    h = torch.zeros(nbytes // 4, dtype=torch.float32).pin_memory()
    d = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for k in range(3):
        g1.replay() if k % 2 == 0 else g2.replay()
    ts = []
    for rep in range(3):
        a = time.perf_counter(); d.copy_(h, non_blocking=True); ts.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    print(f"  {nbytes:8d} B: host ms per copy " + " ".join(f"{t:.2f}" for t in ts))
