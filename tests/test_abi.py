"""CPU: the C-ABI library loads and exports every symbol include/octa_hip.h declares; argument
checking returns error codes (no compute needs a GPU here)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "octave_amd", "libocta_hip.so")):
        g.build()
    from octave_amd._lib import lib
    return lib()


def test_header_declares_the_whole_surface():
    from octave_amd._lib import parse_header
    sig = parse_header()
    for name in ("octa_conv2d_fwd", "octa_conv2d_dgrad", "octa_conv2d_wgrad", "octa_bn_stats", "octa_bn_apply", "octa_bn_bwd",
                 "octa_maxpool3s2_fwd", "octa_avgpool_fwd", "octa_splat_gap", "octa_splat_apply", "octa_splat_bwd", "octa_aag_fwd",
                 "octa_aag_bwd", "octa_wpce_dice_fwd", "octa_wpce_dice_bwd", "octa_interlayer_kl_fwd", "octa_interlayer_kl_bwd",
                 "octa_lsgan_fwd", "octa_noise_clip_fwd", "octa_spectral_norm_fwd", "octa_fullconv_fwd", "octa_adam_step",
                 "octa_version", "octa_last_error"):
        assert name in sig, name
    assert len(sig) >= 44


def test_library_exports_every_declared_symbol(L):
    dll = ctypes.CDLL(os.path.join(ROOT, "octave_amd", "libocta_hip.so"))
    for name in L.signatures:
        assert hasattr(dll, name), f"{name} declared in octa_hip.h but not exported"
    assert L.octa_version() >= 301


def test_bad_arguments_return_error_codes_not_crashes(L):
    from octave_amd._lib import ConvDesc, OctaError
    d = ConvDesc()
    with pytest.raises(OctaError, match="dtype|dims|null"):
        L.octa_conv2d_fwd(ctypes.byref(d), None, None, None, None, None)
    with pytest.raises(OctaError):
        L.octa_bn_stats(None, 0, 0, 0, 0, 0, 1e-5, 0.1, None, None, None, None, None, None)
    with pytest.raises(OctaError, match="num_classes|null"):
        L.octa_aag_fwd(None, None, None, None, None, 1, 1, 8, 9, 0, 0, None)
    assert L.raw("octa_last_error")() is not None


def test_cpu_tensor_is_rejected_loudly(L):
    import torch
    from octave_amd import functional as F_
    from octave_amd._lib import OctaError
    with pytest.raises(OctaError, match="no CPU fallback"):
        F_.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 1, 1))


def test_missing_library_fails_loudly(monkeypatch):
    from octave_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libocta_hip.so")
    with pytest.raises(_lib.OctaError, match="no fallback"):
        _lib._Lib()


def test_module_state_dict_matches_reference_layout():
    """Keys, shapes and order of OctaScribbleNet.state_dict() equal the reference's (derived in oracle/shapes.py)."""
    import torch
    from architectures.models.octa import OctaScribbleNet
    from oracle.shapes import octa_state_shapes
    net = OctaScribbleNet(torch.Size((2, 3, 48, 48)), torch.Size((2, 2, 48, 48)), True, False)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    want = octa_state_shapes(48)
    assert mine == dict(want)
    # ... and equal to the key / shape list dumped from the reference itself (tests/golden/extras.npz, oracle/gen_golden.py extras)
    import numpy as np, os
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "extras.npz"), allow_pickle=False) as z:
        assert list(mine.keys()) == z["layout/octa/keys"].tolist()
        assert [",".join(str(v) for v in s) for s in mine.values()] == z["layout/octa/shapes"].tolist()
    assert sum(p.numel() for p in net.segmentor.parameters()) == 73056784
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1))


def test_group_merge_factor_host_logic():
    """functional._densify: which grouped 3x3 layers run with merged groups, and with how many per dense block (host logic only)."""
    import torch
    from octave_amd import functional as F_
    bf = torch.bfloat16
    assert F_._densify(4, 32, 64, 3, 3, 1, 1, 400, 400, bf) == 4        # decoder_0: 8 -> 16 per group: one dense conv
    assert F_._densify(4, 64, 128, 3, 3, 1, 1, 200, 200, bf) == 2       # decoder_1: 16 -> 32 per group: pairs (two groups of 32 -> 64)
    assert F_._densify(4, 64, 128, 3, 3, 1, 1, 200, 200, torch.float32) == 0      # fp32 keeps the grouped kernels
    assert F_._densify(2, 64, 128, 3, 3, 1, 1, 100, 100, bf) == 0       # encoder_2: 32 -> 64 per group is already a resident-weight shape
    assert F_._densify(4, 64, 128, 3, 3, 1, 1, 100, 100, bf) == 0       # small images stay grouped
    assert F_._densify(1, 64, 128, 3, 3, 1, 1, 200, 200, bf) == 0
    assert F_._densify(4, 64, 128, 1, 1, 1, 0, 200, 200, bf) == 0       # 3x3 / stride 1 / pad 1 only
    assert F_._densify(4, 64, 128, 3, 3, 2, 1, 200, 200, bf) == 0


def test_fanout_offer_host_logic():
    """functional.offer_fanout / take_fanout / skip_with_fanout / _take_parked (round 5, host logic only): an offer is made only for a
    differentiable GPU tensor, taken at most once, never inside a side branch, and the decoder side stashes only when it was taken."""
    import torch
    from octave_amd import functional as F_
    x = torch.zeros(1, 8, 2, 2, requires_grad=True)
    assert F_.offer_fanout(x) is None and F_.take_fanout() is None          # CPU tensor: nothing to fuse (and no stale offer left)
    h = F_.GradHolder()
    F_._FANOUT_TLS.offer = h
    assert F_.take_fanout() is h and h.taken and F_.take_fanout() is None  # taken once
    h2 = F_.GradHolder()
    F_._FANOUT_TLS.offer = h2
    old = F_._IN_SIDE
    F_._IN_SIDE = True
    try:
        assert F_.take_fanout() is None and not h2.taken                    # a pool on a side stream leaves the sum to autograd
    finally:
        F_._IN_SIDE = old
    assert F_.skip_with_fanout(x, h2) is x and F_.skip_with_fanout(x, None) is x
    y = F_.skip_with_fanout(x, h)                                           # taken -> the gradient is parked, not returned
    assert y is not x and type(y.grad_fn).__name__ == "StashGradFnBackward"
    y.sum().backward()
    assert x.grad is None and h.grad is not None and float(h.grad.sum()) == 32.0
    g = F_._take_parked(h)
    assert h.consumed and h.grad is None and float(g.sum()) == 32.0 and F_._take_parked(None) is None
    F_._FANOUT_TLS.offer = h2
    F_.withdraw_fanout()
    assert F_.take_fanout() is None


def test_upsampling_cat_after_respects_module_hooks():
    """Upsampling.cat_after (round 5) bypasses nn.Module.__call__ for the fused path: with a hook on the module (or on its transposed conv)
    it must call the module, so that the hook sees the module's own output (here: the call reaches the op, which refuses CPU tensors)."""
    import pytest
    import torch
    from architectures.extra.resnest import Upsampling
    from octave_amd._lib import OctaError
    u = Upsampling(16, 8)
    seen = []
    u.register_forward_pre_hook(lambda m, i: seen.append("pre"))
    with pytest.raises(OctaError):
        u.cat_after(torch.zeros(1, 8, 4, 4), torch.zeros(1, 16, 2, 2))
    assert seen == ["pre"]                                                  # the module was called: the separate-ops path
    v = Upsampling(16, 8)
    with pytest.raises(OctaError):
        v.cat_after(torch.zeros(1, 8, 4, 4), torch.zeros(1, 16, 2, 2))     # no hook: functional.upsample_cat, same refusal on the CPU


def test_scratch_argument_checks(L):
    """The per-call scratch (octa_conv_desc.ws, octa_conv2d_wgrad_batch's ws): a misaligned or negative-size buffer is refused before any
    launch; NULL / 0 = none.  (No registration entry points exist any more: the library keeps no device pointer between calls.)"""
    from octave_amd._lib import OctaError, WgradJob
    assert not hasattr(L._dll, "octa_conv_splitk_workspace") and not hasattr(L._dll, "octa_wgrad_fold_workspace")
    jobs = (WgradJob * 1)()
    L.octa_conv2d_wgrad_batch(jobs, 0, None, 0, None)         # empty batch, no scratch: nothing launched
    with pytest.raises(OctaError, match="aligned"):
        L.octa_conv2d_wgrad_batch(jobs, 0, ctypes.c_void_p(8), 1 << 20, None)
    with pytest.raises(OctaError, match="aligned|NULL"):
        L.octa_conv2d_wgrad_batch(jobs, 0, None, 1 << 20, None)


def test_no_static_device_pointer_in_csrc():
    """SURVEY 8(b) Ownership: the library never retains a device pointer past return.  Static check: no file-scope pointer variable in
    csrc/ (thread-local call-scoped session state and host-side strings excepted)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octave_amd", "csrc")
    bad = []
    for fn in sorted(os.listdir(root)):
        if not fn.endswith((".hip", ".hpp", ".cpp")):
            continue
        for i, line in enumerate(open(os.path.join(root, fn)), 1):
            if re.match(r"^static\s+(?!thread_local)(?!const\s+char)(?!inline)(?!bool\b)(?!int\b)[\w:<> ]*\*\s*g_\w+", line):
                bad.append(f"{fn}:{i}: {line.strip()}")
    assert not bad, "file-scope device pointer(s): " + "; ".join(bad)


def test_bench_rank0_block_issues_no_training_step():
    """bench.py with N > 1: whatever runs on rank 0 ALONE must not run a training step (its gradient exchange would have no partners
    and the run would hang -- the round-4 bug of the roofline leg).  Static check of main()'s `if rank == 0:` block."""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    blocks = [n for n in ast.walk(main) if isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and isinstance(n.test.left, ast.Name)
              and n.test.left.id == "rank" and isinstance(n.test.ops[0], ast.Eq) and getattr(n.test.comparators[0], "value", None) == 0]
    assert blocks, "bench.py: no `if rank == 0:` block found in main()"
    forbidden = {"step", "roofline_record", "autotune_launch", "comm_diagnosis", "capture"}
    for blk in blocks:
        for node in ast.walk(blk):
            if isinstance(node, ast.Call):
                f = node.func
                name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else "")
                assert name not in forbidden, f"bench.py line {node.lineno}: `{name}(...)` inside a rank-0-only block would leave its collectives without partners"
