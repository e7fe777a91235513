"""CPU, world_size 2 over gloo: the data-parallel plumbing of octave_amd.train (flat gradient arena,
all-reduce, 1/world scaling) with the oracle standing in for the per-rank compute.  Semantics under
test: per-replica BatchNorm/statistics, gradients AVERAGED over ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_state():
    from oracle.fill import fill_tensor
    from oracle.shapes import discriminator_shapes
    shapes = discriminator_shapes(32, in_ch=2, nf=8, depth=4, prefix="")
    P = {}
    for k, shp in shapes.items():
        t = fill_tensor(k, torch.empty(shp))
        P[k] = torch.nn.Parameter(t) if not k.endswith(("weight_u", "weight_v")) else t
    return P


def _shard(rank, B=2, H=32):
    from oracle.fill import hash_input

    def pyr(seed):
        return [F.softmax(2 * hash_input((B, 2, H >> i, H >> i), seed + i + 100 * rank, -1, 1), dim=1) for i in range(5)]
    return pyr(10), pyr(20)


def _local_grads(P, rank, reset=True):
    from oracle import ref_ops as R
    real, fake = _shard(rank)
    for p in P.values():
        if isinstance(p, torch.nn.Parameter) and reset:
            p.grad = None
    Q = {k: (v if isinstance(v, torch.nn.Parameter) else v.clone()) for k, v in P.items()}   # u/v: same start on every rank
    noise = torch.zeros(32, 32)
    l = R.ls_discriminator_loss(R.discriminator_forward(real, Q, prefix="", noise=noise, flip=False),
                                R.discriminator_forward(fake, Q, prefix="", noise=noise, flip=False))
    l.backward()
    return {k: p.grad.clone() for k, p in P.items() if isinstance(p, torch.nn.Parameter)}


def _worker(rank, world, port, q):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    ok = False
    try:
        from octave_amd.train import FlatArena
        P = _make_state()
        named = [(k, p) for k, p in P.items() if isinstance(p, torch.nn.Parameter)]
        params = [p for _, p in named]
        # three gradient groups in "completion order" (as TrainStep lays the segmentor out): buckets must tile the arena in order
        third = (len(named) + 2) // 3
        groups = [(f"g{i}", [k for k, _ in named[i * third:(i + 1) * third]]) for i in range(3)]
        arena = FlatArena(named, groups, min_bucket=1)
        assert [b[0] for b in arena.buckets] == ["g0", "g1", "g2"] and arena.buckets[0][1] == 0 and arena.buckets[-1][2] == arena.numel
        assert all(a[2] == b[1] for a, b in zip(arena.buckets, arena.buckets[1:]))
        merged = FlatArena([(k, torch.nn.Parameter(p.detach().clone())) for k, p in named], groups, min_bucket=1 << 30)
        assert len(merged.buckets) == 1 and merged.buckets[0][1:] == (0, merged.numel)        # small groups merge forward
        # replicas: rank 1 starts from different parameters; broadcast() makes them rank 0's (DistributedDataParallel's contract)
        ref0 = arena.p.clone()
        if rank == 1:
            with torch.no_grad():
                arena.p.add_(1.0)
        arena.broadcast(0)
        chk = arena.p.clone()
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        assert torch.equal(chk, world * arena.p) and torch.equal(arena.p, ref0)
        assert all(p.grad.data_ptr() >= arena.g.data_ptr() for p in params)
        arena.zero_grad()
        g_local = _local_grads(P, rank, reset=False)
        # the oracle's autograd wrote into the arena views in place (p.grad pre-assigned)
        for (k, p) in [(k, p) for k, p in P.items() if isinstance(p, torch.nn.Parameter)]:
            assert p.grad.data_ptr() >= arena.g.data_ptr() and torch.equal(p.grad, g_local[k])
        for _, lo, hi in arena.buckets:          # the bucket schedule TrainStep walks (one collective per bucket, in arena order)
            arena._reduce_range(lo, hi, None)
        sd = arena.state_dict()
        arena.load_state_dict(sd)
        assert set(sd["exp_avg"]) == {k for k, _ in named} and sd["step"] == 0
        avg = {k: p.grad.clone() / world for k, p in P.items() if isinstance(p, torch.nn.Parameter)}
        if rank == 0:
            P2 = _make_state()
            want = None
            for r in range(world):
                g = _local_grads(P2, r)
                want = g if want is None else {k: want[k] + g[k] for k in g}
            err = max(((avg[k] - want[k] / world).abs().max().item()) for k in want)
            q.put(("ok", err, arena.numel, len(params)))
        ok = True
    except Exception as e:   # surface the failure to the parent
        import traceback
        q.put(("fail", traceback.format_exc(), 0, 0))
        raise
    finally:
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def test_flat_arena_all_reduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        status, err, numel, nparams = q.get(timeout=200)
    finally:
        for p in procs:
            p.join(timeout=100)
            if p.is_alive():
                p.kill()
    assert status == "ok", err
    assert err < 1e-6, err
    assert numel >= 1 and nparams == 20


def _worker_buckets_bf16(rank, world, port, q):
    """Two TrainStep-shaped ranks: the segmentor-style arena (groups in gradient-completion order), the bucket schedule a step
    walks -- buckets handed over from the stage marks in completion order while `backward` is still producing the later ones,
    the stragglers afterwards with skip=started -- and the reduced-precision exchange (grad_comm_dtype=bfloat16)."""
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    ok = False
    try:
        from octave_amd import train as T
        P = _make_state()
        named = [(k, p) for k, p in P.items() if isinstance(p, torch.nn.Parameter)]
        quarter = (len(named) + 3) // 4
        tags = ["decoder_3", "decoder_4", "encoder_4", "end"]
        groups = [(tags[i], [k for k, _ in named[i * quarter:(i + 1) * quarter]]) for i in range(4)]
        arena = T.FlatArena(named, groups, min_bucket=1)
        assert [b[0] for b in arena.buckets] == tags
        arena.zero_grad()
        g_local = _local_grads(P, rank, reset=False)           # "backward": every rank's own shard, written into the arena views
        tag_to_bucket = {t: i for i, (t, _, _) in enumerate(arena.buckets)}
        started = []
        for tag in ("decoder_3", "decoder_4", "encoder_4"):     # the marks fire in completion order; "end" has no mark
            i = tag_to_bucket[tag]
            arena.all_reduce_bucket_async(world, None, i, torch.bfloat16)
            started.append(i)
        arena.all_reduce_begin(world, None, torch.bfloat16, skip=started)
        arena.all_reduce_end(world, None)
        # every bucket was reduced exactly once: sum over ranks of the bf16-rounded local gradients, rounded to bf16 again
        P2 = _make_state()
        want = None
        for r in range(world):
            g = {k: v.bfloat16().float() for k, v in _local_grads(P2, r).items()}
            want = g if want is None else {k: want[k] + g[k] for k in g}
        err, scale = 0.0, 0.0
        for k, p in named:
            w = want[k].bfloat16().float()
            err = max(err, (p.grad - w).abs().max().item())
            scale = max(scale, w.abs().max().item())
            # and against the exact fp32 average: one rounding per rank + one for the sum
            exact = sum(_local_grads(P2, r)[k] for r in range(world))
            assert (p.grad - exact).abs().max().item() <= 3 * 2.0 ** -8 * exact.abs().max().item() + 1e-12, k
        chk = arena.g.clone()
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        assert torch.equal(chk, world * arena.g), "ranks disagree after the exchange"
        if rank == 0:
            q.put(("ok", err, scale, len(started)))
        ok = True
    except Exception:
        import traceback
        q.put(("fail", traceback.format_exc(), 0, 0))
        raise
    finally:
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def test_bucket_schedule_bf16_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buckets_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        status, err, scale, nstarted = q.get(timeout=200)
    finally:
        for p in procs:
            p.join(timeout=100)
            if p.is_alive():
                p.kill()
    assert status == "ok", err
    assert err <= 2.0 ** -8 * scale and nstarted == 3, (err, scale, nstarted)


def _worker_identical_after_update(rank, world, port, q):
    """Two ranks that start from DIFFERENT parameters and see DIFFERENT shards: after the broadcast, the bucketed all-reduce and
    one averaged Adam update the replicas hold bit-identical parameters, equal to a single process's update with the average of
    both shards' gradients.  (The fused Adam launch is a HIP kernel; on the host the same update is torch.optim.Adam over the
    arena's parameter views with grad_scale = 1 / world applied to the reduced arena, which is what octa_adam_step folds in.)"""
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    ok = False
    try:
        from octave_amd import train as T
        P = _make_state()
        named = [(k, p) for k, p in P.items() if isinstance(p, torch.nn.Parameter)]
        half = (len(named) + 1) // 2
        groups = [("decoder_4", [k for k, _ in named[:half]]), ("end", [k for k, _ in named[half:]])]
        arena = T.FlatArena(named, groups, min_bucket=1)
        start0 = arena.p.clone()
        if rank == 1:
            with torch.no_grad():
                arena.p.mul_(1.5).add_(0.01)            # a replica built from another seed
        arena.broadcast(0)
        assert torch.equal(arena.p, start0) if rank == 0 else True
        arena.zero_grad()
        _local_grads(P, rank, reset=False)              # this rank's shard, gradients land in the arena views
        mine = arena.g.clone()
        arena.all_reduce_bucket_async(world, None, 0, None)
        arena.all_reduce_begin(world, None, None, skip=[0])
        arena.all_reduce_end(world, None)
        with torch.no_grad():
            arena.g.mul_(1.0 / world)                   # grad_scale of the fused Adam launch
        opt = torch.optim.Adam([p for _, p in named], lr=1e-3)
        opt.step()
        chk = arena.p.clone()
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        same = torch.equal(chk, world * arena.p)        # both ranks hold the same bits (x2 is exact)
        moved = (arena.p - start0).abs().max().item() if rank == 0 else 0.0
        other = mine.clone()
        dist.all_reduce(other, op=dist.ReduceOp.SUM)
        differ = not torch.equal(other, world * mine)   # the shards really produced different local gradients
        if rank == 0:
            # single-process reference: the average of both shards' gradients, one Adam step from rank 0's parameters
            P2 = _make_state()
            named2 = [(k, p) for k, p in P2.items() if isinstance(p, torch.nn.Parameter)]
            g = [_local_grads(P2, r) for r in range(world)]
            for k, p in named2:
                p.grad = sum(gr[k] for gr in g) / world
            torch.optim.Adam([p for _, p in named2], lr=1e-3).step()
            err = max((p.detach() - dict(named)[k].detach()).abs().max().item() for k, p in named2)
            q.put(("ok", (same, differ, moved, err), 0, 0))
        ok = True
    except Exception:
        import traceback
        q.put(("fail", traceback.format_exc(), 0, 0))
        raise
    finally:
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def test_replicas_identical_after_broadcast_and_averaged_update_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_identical_after_update, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        status, res, _, _ = q.get(timeout=200)
    finally:
        for p in procs:
            p.join(timeout=100)
            if p.is_alive():
                p.kill()
    assert status == "ok", res
    same, differ, moved, err = res
    assert same and differ and moved > 1e-4 and err < 1e-6, res


def test_bench_shards_data_by_rank():
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.modules["bench"] = bench
    spec.loader.exec_module(bench)
    x0, ys0, r0 = bench.synth_batch(2, 32, 0, "cpu")
    x1, ys1, r1 = bench.synth_batch(2, 32, 1, "cpu")
    assert x0.shape == (2, 3, 32, 32) and ys0.shape == (2, 2, 32, 32) and r0.shape == (2, 2, 32, 32)
    assert not torch.equal(x0, x1) and not torch.equal(ys0, ys1)
    assert torch.equal(x0[:, 0], x0[:, 1]) and float(ys0.sum(1).max()) <= 1.0 and torch.all(r0.sum(1) == 1)
    assert 1 <= bench.host_cores() <= 16
    d = type("D", (), dict(upshuffle=0, B=2, OH=4, OW=4, Cout=8, Cin=16, groups=2, KH=3, KW=3, H=4, W=4))()
    assert bench.conv_flops(d) == 2.0 * 2 * 16 * 8 * 8 * 9
