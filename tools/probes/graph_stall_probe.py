"""What makes queued hipGraph replays stall on this stack?  Variants of a synthetic 'step' (several graphs, many nodes, memset /
memcpy nodes, eager kernels in between), replayed 12x back to back; device time between step-end events and host time per step."""
import sys, time, torch
dev = torch.device("cuda:0")
x = torch.randn(8 << 20, device=dev)           # 32 MB: ~12 us per elementwise kernel
y = torch.empty_like(x)
z = torch.zeros(1 << 16, device=dev)


def capture(nodes, memnodes):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.mul(x, 1.0001, out=y)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for i in range(nodes):
            torch.mul(x, 1.0001, out=y)
            if memnodes and i % 16 == 0:
                z.zero_()                       # memset node
                y[:1 << 16].copy_(z)            # D2D memcpy node
    return g


def run(label, graphs, eager_between, n=12):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ts = []
    t0 = time.perf_counter(); evs[0].record()
    for k in range(n):
        a = time.perf_counter()
        for g in graphs:
            if eager_between:
                z.add_(1.0)
            g.replay()
        evs[k + 1].record()
        ts.append((time.perf_counter() - a) * 1e3)
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    print(f"{label:46s} total {tot:6.1f} ms/step host {host:5.1f} | device ms: " + " ".join(f"{evs[k].elapsed_time(evs[k + 1]):.0f}" for k in range(n))
          + " | host ms: " + " ".join(f"{t:.0f}" for t in ts), flush=True)


g300 = capture(300, False)
run("1 graph x 300 nodes", [g300], False)
g1300 = capture(1300, False)
run("1 graph x 1300 nodes", [g1300], False)
g2600 = capture(2600, False)
run("1 graph x 2600 nodes", [g2600], False)
four = [capture(700, False), capture(400, False), capture(100, False), capture(100, False)]
run("4 graphs (700/400/100/100)", four, False)
run("4 graphs + eager kernel before each", four, True)
gm = capture(1300, True)
run("1 graph x 1300 nodes + memset/memcpy nodes", [gm], False)
fourm = [capture(700, True), capture(400, True), capture(100, True), capture(100, True)]
run("4 graphs with memset/memcpy nodes + eager", fourm, True)
