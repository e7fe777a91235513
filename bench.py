#!/usr/bin/env python3
"""bench.py -- images/sec of the OCTAve segmentor+discriminator training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the config its metric is quoted on): OctaScribbleNet, full
adversarial loop (segmentor fwd/bwd + WeightedPartialCE + Dice + InterlayerDivergence + LS-GAN
generator term, Adam; then the LS-GAN discriminator step, Adam), batch 16 per GPU, 400x400, bf16
activations, synthetic OCTA-like data, default-initialised weights.  Weak scaling: every rank runs
the same per-GPU batch; gradients are all-reduced over RCCL.

One JSON line is printed by rank 0.  Extra objects: "roofline" (MFMA-bound conv engine kernel:
algorithmic FLOPs / HIP-event time of its launches), "cpu_baseline" (the CPU oracle's step on
the host cores, a bounded sample), "sustained_ms_per_step" (>= 500 back-to-back steps), "dice_vs_ref"
(eval-mode one-hot masks against the reference's own, tests/golden/round4.npz) and, with more than
one rank, "dist" (ranks the RCCL communicator spans, per-bucket all-reduce times, exposed wait).
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16 / f16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0         # HBM3E, MI355X_MICROARCH.md


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask, cgroup quota, capped at 16 (the GPU
    box gives one-GPU jobs a 16-CPU share; oversubscribing it makes the baseline meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("OCTA_CPU_THREADS", "16"))))


def visible_gpus():
    """GPUs this process would see, WITHOUT touching the HIP runtime: KFD topology nodes with SIMDs (CPUs have simd_count 0),
    narrowed by ROCR_/HIP_/CUDA_VISIBLE_DEVICES.  None when sysfs does not say (then --gpus is trusted and a rank that finds no
    device fails loudly)."""
    import glob
    n = 0
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    for path in paths:
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def synth_batch(B, H, rank, device):
    """grayscale OCTA-like plane replicated to 3 channels; scribbles ~5 % per class;
    dense real vessel mask, one-hot.  On the GPU the batch is generated ON the device (octa_synth_octa, a counter-based
    generator keyed on (seed, element)); the host generator below serves the CPU baseline leg and the CPU tests."""
    if torch.device(device).type == "cuda":
        from octave_amd import functional as F_
        return F_.synth_octa_batch(B, H, H, seed=1234 + 7919 * rank, device=device)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand((B, 1, H, H), generator=g).repeat(1, 3, 1, 1)
    g2 = torch.Generator().manual_seed(4321 + rank)
    u = torch.rand((B, 1, H, H), generator=g2)
    ys = torch.zeros(B, 2, H, H)
    ys[:, 1:2] = (u < 0.05).float()
    ys[:, 0:1] = ((u > 0.5) & (u < 0.55)).float()
    g3 = torch.Generator().manual_seed(999 + rank)
    dense = (torch.rand((B, H, H), generator=g3) > 0.8).long()
    real = torch.nn.functional.one_hot(dense, 2).permute(0, 3, 1, 2).float().contiguous()
    return x.to(device), ys.to(device), real.to(device)


def conv_bytes(d, kind):
    """Algorithmic HBM bytes of one conv-engine launch: every operand once (activations in the compute dtype, the weight
    gradient in fp32)."""
    es = 4 if d.dtype == 0 else 2
    groups = getattr(d, "alg_groups", d.groups)
    wel = d.Cout * (d.Cin // groups) * d.KH * d.KW
    xin = d.B * d.H * d.W * d.Cin
    yout = d.B * d.OH * d.OW * (d.Cout if not d.upshuffle else d.Cout)
    if kind == "wgrad":
        return es * (xin + yout) + 4.0 * wel
    return es * (xin + yout + wel)


def conv_flops(d):
    if d.upshuffle:
        return 2.0 * d.B * d.H * d.W * d.Cin * d.Cout
    return 2.0 * d.B * d.OH * d.OW * d.Cout * (d.Cin // getattr(d, "alg_groups", d.groups)) * d.KH * d.KW    # algorithmic (densified layers too)


def roofline_record(step, batch):
    """Record the conv-engine launches of one EAGER step.  With more than one rank EVERY rank must call this: the step it runs
    exchanges gradients like any other (a recording on rank 0 alone would leave its all-reduces without partners)."""
    from octave_amd import functional as F_
    graphs, step._graphs = step._graphs, None      # record an EAGER step (a graph replay launches nothing from Python)
    F_.start_recording()
    try:
        step(*batch)
        rec = F_.stop_recording()
    finally:
        step._graphs = graphs
    torch.cuda.synchronize()
    return rec


def roofline_leg(step, rec, batch, dtype_name):
    """Replay the recorded conv-engine launches once, in order, each between two HIP events on the stream it is launched on
    (torch's current stream), and aggregate per kernel instance (the name is the template instance the library reports it
    dispatched: octa_last_conv_kernel).  No collective in here: rank 0 runs it alone."""
    import ctypes
    from octave_amd import functional as F_
    from octave_amd._lib import lib
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    agg = {}
    pending = []
    # as during the segmentor phase of a step: the recorded descriptors carry the tail-split scratch of their call (octa_conv_desc.ws);
    # the batched weight gradients get the phase's fold scratch as an argument, as in the step
    fold_ws = getattr(step, "_fold_ws", None)
    fold_args = (None, 0) if fold_ws is None else (fold_ws.data_ptr(), fold_ws.numel() * 4)
    evict = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    dw_scratch = {}
    for kind, d, ptrs, keep in rec:
        def launch():
            if kind == "wgrad_batch":           # (job array, count): one batched weight-gradient flush, re-run into the same gradient slots
                L.octa_conv2d_wgrad_batch(d, ptrs, fold_args[0], fold_args[1], st)
            elif kind == "fwd":
                L.octa_conv2d_fwd(ctypes.byref(d), ptrs[0], ptrs[1], ptrs[2], ptrs[3], st)
            elif kind == "dgrad":
                L.octa_conv2d_dgrad(ctypes.byref(d), ptrs[0], ptrs[1], ptrs[2], st)
            else:
                shape, stride = ptrs[2], ptrs[3]
                key = id(d)                    # a gradient buffer of its own per launch, like the arena slots in the step
                if key not in dw_scratch:
                    n = sum((s - 1) * t for s, t in zip(shape, stride)) + 1
                    dw_scratch[key] = torch.zeros(n, dtype=torch.float32, device="cuda")
                L.octa_conv2d_wgrad(ctypes.byref(d), ptrs[0], ptrs[1], dw_scratch[key].data_ptr(), (ctypes.c_int64 * 4)(*stride), None, st)
        # ONE launch per recorded call, in program order, without a warm-up launch and behind a 512 MB sweep that evicts L2 and
        # the 256 MB memory-side cache: inside the step ~10 other kernels run between two conv launches, so the operands come
        # from HBM (back-to-back conv launches alone read them out of the cache and time 10-15 % faster than in the trace)
        evict.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        fl = sum(conv_flops(d[i].d) for i in range(ptrs)) if kind == "wgrad_batch" else conv_flops(d)
        by = sum(conv_bytes(d[i].d, "wgrad") for i in range(ptrs)) if kind == "wgrad_batch" else conv_bytes(d, kind)
        pending.append((e0, e1, L.octa_last_conv_kernel().decode().split("+tail")[0].split("+fold")[0], fl, by))      # "+tailNxP" / "+fold": the same kernel with its tail split / partial tiles (incl. the fix-up / fold launch)
    torch.cuda.synchronize()
    peak = PEAK_BF16_TFLOPS if dtype_name in ("bf16", "f16") else PEAK_F32_TFLOPS
    for e0, e1, kname, fl, by in pending:
        ms = e0.elapsed_time(e1)
        a = agg.setdefault(kname, [0.0, 0.0, 0, 0.0, 0.0, 0.0])
        a[0] += fl
        a[1] += ms * 1e-3
        a[2] += 1
        a[3] += by
        # the roofline of ONE launch: the slower of its MFMA time and its HBM time
        a[4] += fl / (peak * 1e12)
        a[5] += by / (PEAK_HBM_GBS * 1e9)
    tot_f = sum(a[0] for a in agg.values())
    tot_t = sum(a[1] for a in agg.values())
    name, (f, t, n, byt, t_mfma, t_hbm) = max(agg.items(), key=lambda kv: kv[1][1])
    # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # their own runs, gfx950 correction applied by tools/pmc_traffic.py); null when the kernel was not profiled
    traffic, traffic_src = pmc_traffic(name, batch)
    per_kernel = {k: {"launches_per_step": v[2], "avg_us": round(v[1] / v[2] * 1e6, 2), "tflops": round(v[0] / v[1] / 1e12, 2),
                      "gbs": round(v[3] / v[1] / 1e9, 1), "bound": "mfma" if v[4] >= v[5] else "hbm",
                      "frac": round(max(v[4], v[5]) / v[1], 4)} for k, v in agg.items()}
    # which roofline bounds the dominant kernel: summed over its launches, the larger of the ideal MFMA time (algorithmic FLOPs
    # at the dense peak) and the ideal HBM time (algorithmic bytes at 8 TB/s); `achieved` is quoted in that roofline's unit
    if t_mfma >= t_hbm:
        bound = {"bound": "mfma", "achieved": round(f / t / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(f / t / 1e12 / peak, 4)}
    else:
        bound = {"bound": "hbm", "achieved": round(byt / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(byt / t / 1e9 / PEAK_HBM_GBS, 4)}
    return {
        **bound, "kernel": name, "traffic": traffic, "traffic_source": traffic_src, "mfma_frac": round(f / t / 1e12 / peak, 4), "hbm_frac": round(byt / t / 1e9 / PEAK_HBM_GBS, 4),
        "launches_per_step": n, "avg_launch_us": round(t / n * 1e6, 2), "flops_per_launch_avg": f / n, "bytes_per_launch_avg": byt / n,
        "all_conv_kernels": {"achieved": round(tot_f / tot_t / 1e12, 2), "frac": round(tot_f / tot_t / 1e12 / peak, 4),
                             "time_ms_per_step": round(tot_t * 1e3, 3), "gflop_per_step": round(tot_f / 1e9, 1)},
        "per_kernel": per_kernel,
    }


def dice_vs_ref_leg(dev, sizes=(400, 304)):
    """BASELINE.json's metric ends "; Dice vs ref": ``ResnestUNet.predict(x, 'one-hot')`` (segmentor/compose.py:189-199) in eval
    mode on the HIP path, on the INPUT of the reference fixture tests/golden/round4.npz (closed-form weights and image,
    octave_amd/synth.py; B = 2), against the mask the reference itself produced there (oracle/gen_golden.py round4).  Dice
    coefficient per class (segmentor/losses.py:70-74 without the 1 -), the minimum over samples and classes is the figure; plus
    the number of pixels that differ and how many of those are decidable in float64 (must be 0)."""
    import numpy as np
    from architectures.models.octa import OctaScribbleNet
    from octave_amd import functional as F_
    from octave_amd.synth import fill_state_dict, hash_input
    G = np.load(os.path.join(ROOT, "tests", "golden", "round4.npz"))
    out = {}
    for H in sizes:
        B = int(G[f"eval{H}/shape"][0])
        net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), False, False)
        fill_state_dict(net.state_dict())
        net = net.to(dev).eval()
        x = hash_input((B, 1, H, H), 1234).repeat(1, 3, 1, 1).to(dev)
        with torch.no_grad():
            oh = net.segmentor.predict(x, "one-hot")[1]
        n = B * H * H
        want = np.unpackbits(G[f"eval{H}/onehot_cls1_bits"])[:n].reshape(B, H, H).astype(bool)
        ok = np.unpackbits(G[f"eval{H}/decidable_bits"])[:n].reshape(B, H, H).astype(bool)
        got = oh[:, 1].cpu().numpy().astype(bool)
        ref = torch.from_numpy(np.stack([~want, want], 1).astype(np.float32)).to(dev)
        got2 = oh.float()
        inter = (got2 * ref).sum(dim=(2, 3))
        dice_c = (2 * inter / (got2.sum(dim=(2, 3)) + ref.sum(dim=(2, 3)) + 1e-12)).cpu().numpy()      # [B, class]
        out[str(H)] = {"dice_min_over_samples_and_classes": round(float(dice_c.min()), 6),
                       "dice_all_pixels": round(float(F_.dice_coefficient(got2, ref).min()), 6),
                       "pixels": n, "differ": int((got != want).sum()), "differ_decidable_in_f64": int((got != want)[ok].sum())}
        del net
    return {"value": min(v["dice_min_over_samples_and_classes"] for v in out.values()), "mode": "eval, predict('one-hot'), fp32, B=2",
            "fixture": "tests/golden/round4.npz (reference masks, oracle/gen_golden.py round4)", "per_size": out}


def pmc_traffic(name, batch):
    """HBM bytes per launch of kernel `name` from the newest committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    their own runs, gfx950 correction applied by tools/pmc_traffic.py), with the file and the commit it was measured at so
    that the constant is traceable; (None, None) when the kernel was not profiled or the workload is not the profiled one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pm = json.load(open(path))
            k = pm["kernels"]
            if name in k and batch[0].shape[0] == 16 and batch[0].shape[-1] == 400:
                return round(k[name]["hbm_bytes_per_launch"]), {"file": os.path.relpath(path, ROOT), "commit": pm.get("commit"), "measured": pm.get("date")}
        except Exception:
            continue
    return None, None


def cpu_baseline_leg(net_state, seconds_budget=25.0):
    """the CPU oracle (a port of the reference's path, oracle/ref_ops.py) running the full
    adversarial step of BASELINE config 1: batch 2, 304x304, fp32, all 4 losses, Adam, on the host cores."""
    from oracle import ref_ops as R
    ncpu = host_cores()
    torch.set_num_threads(ncpu)
    B, H = 2, 304
    x, ys, real = [t.cpu() for t in synth_batch(B, H, 0, "cpu")]
    P = {}
    for k, v in net_state.items():
        v = v.detach().cpu().clone().contiguous()
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")) and "linear_head_" not in k:
            v.requires_grad_(True)
        P[k] = v
    # the discriminator head depends on the resolution (blocks.py:68-72): re-shape it for 304
    hw = H // 32
    P["discriminator.out.0.weight"] = (torch.randn(1, P["discriminator.out.0.weight"].shape[1], hw, hw) * 0.01).requires_grad_(True)
    seg = [v for k, v in P.items() if k.startswith("segmentor.") and v.requires_grad]
    dis = [v for k, v in P.items() if k.startswith("discriminator.") and v.requires_grad]
    opt_s, opt_d = torch.optim.Adam(seg, lr=1e-4), torch.optim.Adam(dis, lr=1e-4)

    def one():
        opt_s.zero_grad(); opt_d.zero_grad()
        l, att, _ = R.segmentor_loss(P, x, ys)
        l.backward(); opt_s.step()
        opt_d.zero_grad()
        ld = R.discriminator_loss(P, R.mask_pyramid(real), att)
        ld.backward(); opt_d.step()
    one()                                    # warm
    times = []
    t_start = time.time()
    while len(times) < 5 and (time.time() - t_start) < seconds_budget:
        t0 = time.time(); one(); times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": ncpu, "kind": "port",
            "sample": f"{len(times)} full adversarial steps, batch 2, 304x304, fp32 (BASELINE config 1), median {med:.2f} s/step"}


def main():
    # stdout carries exactly ONE line, the JSON: whatever native libraries print there (RCCL writes a five-line version banner to stdout when
    # its communicator is created) goes to stderr instead -- fd 1 is pointed at fd 2 for the run and the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--loss-scale", default=None, help="static loss scale (default 1; 1024 for f16) or 'dynamic'")
    ap.add_argument("--seg-only", action="store_true", help="BASELINE configs[1]: segmentor-only (WPCE+Dice)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="after capture: replay the hipGraphs, launch the same static step from Python, or time both and pick (default)")
    ap.add_argument("--algo-cache", default=None, help="JSON file with measured per-shape conv kernel choices: loaded if present, written after warm-up")
    ap.add_argument("--grad-comm", default="auto", choices=["auto", "f32", "bf16"],
                    help="dtype of the gradient all-reduce: auto = f32 (what DistributedDataParallel on the reference exchanges; 286 MB per step); bf16 halves the bytes over xGMI at two extra roundings per gradient")
    ap.add_argument("--sustained", type=int, default=500, help="replayed steps of the sustained-rate leg after the timed region (0: skip)")
    ap.add_argument("--no-dice", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    t_begin = time.perf_counter()
    if args.gpus > 1 and "RANK" not in os.environ:
        # launched without torchrun: start the N ranks ourselves as a CHILD process and relay its JSON line and exit code.  The
        # launcher itself makes NO HIP call (a process that has opened the GPU must not spawn the ranks on this pool): the
        # devices are counted from the KFD topology in sysfs.  No fallback to fewer ranks.
        import socket
        import subprocess
        ndev = visible_gpus()
        if ndev is not None and ndev < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible; refusing to report a smaller run as n_gpus={args.gpus}")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs the MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("OCTA_DIST_ALWAYS") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # "nccl" is RCCL on ROCm.  NOT bound to the device with device_id=: the eagerly created communicator of that form costs
        # every replayed step +1.2 ms on this stack even when no collective runs (tools/dist_overhead_probe.py, one rank:
        # 25.6 -> 26.9 ms; profiles/r04_dist_overhead.txt); the communicator is created by the first collective instead (the
        # parameter broadcast of TrainStep), and every barrier names its device
        dist.init_process_group(backend="nccl")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the number of ranks must equal --gpus")

    from architectures.models.octa import OctaScribbleNet
    from octave_amd.train import TrainStep, mask_pyramid
    B, H = args.batch, args.size
    torch.manual_seed(0)      # identical default-init weights on every rank
    net = OctaScribbleNet(torch.Size((B, 3, H, H)), torch.Size((B, 2, H, H)), True, False).to(dev).train()
    net_state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()} if (rank == 0 and not args.no_cpu_baseline and world == 1) else None
    cdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    ls = (1024.0 if args.dtype == "f16" else 1.0) if args.loss_scale is None else ("dynamic" if args.loss_scale == "dynamic" else float(args.loss_scale))
    # the reference's DistributedDataParallel exchanges fp32 gradients: the headline does too ("auto" = f32 since round 5; 286 MB per
    # step, ~1.4 ms of ring time at 200 GB/s, overlapped with ~15 ms of backward).  --grad-comm bf16 halves the bytes at two extra
    # roundings per gradient and is reported as what it is in config.grad_comm_dtype
    comm_dt = torch.bfloat16 if args.grad_comm == "bf16" else None
    step = TrainStep(net, lr=1e-4, compute_dtype=cdt, adversarial=not args.seg_only, loss_scale=ls, grad_comm_dtype=comm_dt)
    x, ys, real = synth_batch(B, H, rank, dev)
    batch = (x, ys, mask_pyramid(real))

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_begin:7.1f}s] {msg}", file=sys.stderr, flush=True)
    log(f"model built, {B}x3x{H}x{H} {args.dtype}, world {world}")
    from octave_amd import functional as F_
    if args.algo_cache and F_.load_algo_cache(args.algo_cache):
        log(f"loaded {len(F_._ALGO_CACHE)} measured kernel choices from {args.algo_cache}")
    if not args.no_graph:
        step.capture(*batch)
        log("step captured into hipGraphs (2 eager warm-up steps)")
        if args.launch == "auto":
            step.autotune_launch(*batch)
            log(f"launch path: {step.launch} (ms/step graph {step.launch_timing['graph'] * 1e3:.1f}, eager {step.launch_timing['eager'] * 1e3:.1f})")
        else:
            step.launch = args.launch
    if args.algo_cache and rank == 0 and not os.path.exists(args.algo_cache):
        F_.save_algo_cache(args.algo_cache)
    for i in range(args.warmup):
        step(*batch)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    from octave_amd import _lib as _l
    if _l.PROFILE is not None and rank == 0:
        print(_l.profile_report(), file=sys.stderr, flush=True)
    if use_dist:
        dist.barrier(device_ids=[local])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(*batch)
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the steps (launch-bound if ~ the total)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(device_ids=[local])
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss_seg"].item())
    if not all(math.isfinite(float(v.item())) for v in out.values()):
        sys.exit(f"bench.py: non-finite losses after the timed steps ({ {k: float(v.item()) for k, v in out.items()} }): the measurement is void")
    # the host's own cost of a step: two steps enqueued into an EMPTY queue (nothing to wait for).  t_enq above is not that number: once
    # the runtime's queue is full the host blocks inside the launch, so over 20-40 steps it approaches the device time (5 ms per step
    # over 20 steps, 20 over 40).  All ranks run it (a step exchanges gradients).
    torch.cuda.synchronize()
    th = time.perf_counter()
    for _ in range(2):
        step(*batch)
    host_ms = (time.perf_counter() - th) / 2 * 1e3
    torch.cuda.synchronize()
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step (host: {host_ms:.2f} ms per step into an empty queue; {t_enq / args.steps * 1e3:.1f} ms per step incl. blocking on the full queue)")
    # sustained rate: >= 500 back-to-back steps on the launch path chosen above, nothing on the host inside the loop but the
    # step call itself (no health checks, no .item()); same barriers and max over ranks as the headline
    sustained = None
    if args.sustained > 0:
        if use_dist:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.sustained):
            out = step(*batch)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local])
        ds = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ds = float(t.item())
        sustained = (ds / args.sustained * 1e3, all(math.isfinite(float(v.item())) for v in out.values()))
        log(f"sustained leg done: {sustained[0]:.2f} ms/step over {args.sustained} steps")
    dist_diag = step.comm_diagnosis(*batch) if use_dist else None
    rec = roofline_record(step, batch) if not args.no_roofline else None        # every rank: the recorded step exchanges gradients
    if _l.PROFILE is not None and rank == 0:
        print(_l.profile_report(), file=sys.stderr, flush=True)

    if rank == 0:
        res = {
            "metric": f"images/sec (seg+disc train step) {H}x{H} {args.dtype}" if not args.seg_only else f"images/sec (segmentor-only train step) {H}x{H} {args.dtype}",
            "value": round(world * B * args.steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("OctaScribbleNet full adversarial step (segmentor + LS-GAN discriminator + InterlayerDivergence), "
                                    if not args.seg_only else "OctaScribbleNet segmentor-only step (WeightedPartialCE + Dice), ")
                       + f"batch {B}/GPU, {H}x{H}", "global_batch": world * B, "image": H, "parallelism": f"dp{world}",
                       "weights": "default init, torch.manual_seed(0)", "optimizer": "Adam (fused, flat arena)",
                       "data": "generated on the device (octa_synth_octa)", "grad_buckets": len(step.seg_arena.buckets),
                       "launch": "eager" if args.no_graph else ("hipGraph replay (forward | backward pieces | discriminator step on a second stream | Adam, around the 2 gradient all-reduces)" if step.launch == "graph"
                                                               else "eager launches of the captured static step (auto-tuned: faster than graph replay on this host)")},
            "final_loss_seg": round(loss, 5),
            "host_ms_per_step": round(host_ms, 3),      # host time to enqueue one step into an empty queue (the replay is device-bound while this < ms_per_step)
        }
        res["config"]["grad_comm_dtype"] = "bf16" if comm_dt is not None else "fp32"
        if sustained is not None:
            res["sustained_ms_per_step"] = round(sustained[0], 3)
            res["sustained"] = {"steps": args.sustained, "images_per_sec": round(world * B / sustained[0] * 1e3, 2), "losses_finite": sustained[1],
                                "launch": step.launch}
        if dist_diag is not None:
            res["dist"] = dist_diag
        if not args.no_roofline:
            res["roofline"] = roofline_leg(step, rec, batch, args.dtype)
            log("roofline leg done")
        if net_state is not None:
            res["cpu_baseline"] = cpu_baseline_leg(net_state)
            log("cpu baseline leg done")
        if not args.no_dice:
            step.close()                  # leave the fused-training mode: the Dice leg is a plain eval-mode forward
            res["dice_vs_ref"] = dice_vs_ref_leg(dev)
            log("dice-vs-reference leg done")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
