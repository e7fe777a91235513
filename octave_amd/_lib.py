"""ctypes binding of libocta_hip.so (the C ABI declared in include/octa_hip.h).

The product path has NO fallback: if the library is missing, fails to load, or lacks a symbol
the header declares, importing this module's `lib()` raises.  Signatures are taken from the
header itself so the binding cannot drift from the ABI.
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "octa_hip.h")
LIB_PATH = os.environ.get("OCTA_HIP_LIB", os.path.join(_HERE, "libocta_hip.so"))

OCTA_F32, OCTA_BF16, OCTA_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "H", "W", "OH", "OW", "Cin", "Cout", "KH", "KW", "stride", "pad", "groups",
        "cin_g_pad", "cout_g_pad", "ldx", "xoff", "ldy", "yoff", "dtype", "act", "upshuffle", "algo", "zero_pad")] + \
        [("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_int64)]      # the call's own scratch (tail split / weight-gradient fold)


class WgradJob(ctypes.Structure):
    """octa_wgrad_job: one entry of the batched weight-gradient queue (octa_conv2d_wgrad_batch)."""
    _fields_ = [("d", ConvDesc), ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("dw_strides", ctypes.c_int64 * 4)]


class PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("s_o", ctypes.c_int64), ("s_i", ctypes.c_int64), ("s_h", ctypes.c_int64),
                ("s_w", ctypes.c_int64)] + [(n, ctypes.c_int32) for n in ("kind", "dtype", "Cout_g", "Cin_g", "KH", "KW", "groups", "pad_to")]


class SnJob(ctypes.Structure):
    """octa_sn_job: one layer of a batched spectral normalisation (octa_spectral_norm_fwd_batch)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("w", "u", "v", "sigma", "w_sn", "ws", "uv_saved")] + [("Cout", ctypes.c_int32), ("K", ctypes.c_int32)] + \
               [("packed_fwd", ctypes.c_void_p), ("packed_dgrad_taps", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in ("KH", "KW", "Cin", "pack_dtype")]


class SnBwdJob(ctypes.Structure):
    """octa_sn_bwd_job (octa_spectral_norm_bwd_batch)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("dw_sn", "w_sn", "u", "v", "sigma", "dw", "ws")] + \
               [(n, ctypes.c_int32) for n in ("Cout", "K", "accumulate", "dwsn_khw")]


class OctaError(RuntimeError):
    pass


_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "size_t": ctypes.c_size_t, "octa_stream_t": ctypes.c_void_p,
}


def _ctype_of(decl: str):
    d = decl.strip()
    if "octa_conv_desc" in d and "*" in d:
        return ctypes.POINTER(ConvDesc)
    if "octa_pack_desc" in d and "*" in d:
        return ctypes.c_void_p
    if "*" in d:
        return ctypes.c_void_p
    toks = [t for t in d.replace("const", " ").split() if t]
    base = toks[0]
    if base not in _SCALARS:
        raise OctaError(f"octa_hip.h: cannot map C type in '{decl}'")
    return _SCALARS[base]


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(octa_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        res = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        argtypes = [] if args in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        out[name] = (res, argtypes)
    return out


def header_abi_version(path: str = HEADER) -> int:
    m = re.search(r"#define\s+OCTA_HIP_ABI_VERSION\s+(\d+)", open(path).read())
    if not m:
        raise OctaError("octa_hip.h: OCTA_HIP_ABI_VERSION is not defined")
    return int(m.group(1))


PROFILE = {} if os.environ.get("OCTA_PROFILE") == "1" else None


def profile_report(reset=True):
    rows = sorted(PROFILE.items(), key=lambda kv: -kv[1][0]) if PROFILE else []
    txt = "\n".join(f"{n:32s} {v[0] * 1e3:10.2f} ms {v[1]:6d} calls {v[0] / max(v[1], 1) * 1e6:10.1f} us/call" for n, v in rows)
    if reset and PROFILE is not None:
        PROFILE.clear()
    return txt


_lock = threading.Lock()
_LIB = None


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise OctaError(
                f"libocta_hip.so not found at {LIB_PATH}: build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (or octave_amd/csrc/build.sh). "
                f"There is no fallback path.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both share ONE HIP runtime)
        self._dll = ctypes.CDLL(LIB_PATH)
        self.signatures = parse_header()
        for name, (res, argtypes) in self.signatures.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise OctaError(f"libocta_hip.so lacks symbol {name} declared in octa_hip.h") from e
            fn.restype = res
            fn.argtypes = argtypes
        self._dll.octa_last_error.restype = ctypes.c_char_p
        # a stale build (or a foreign OCTA_HIP_LIB) would be called with mismatched structs and argument lists
        want, got = header_abi_version(), int(self._dll.octa_version())
        if got != want:
            raise OctaError(f"{LIB_PATH} was built for ABI revision {got}, include/octa_hip.h declares {want}: rebuild it (octave_amd/csrc/build.sh)")

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        if PROFILE is not None:
            return self._call_profiled(name, *args)
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            msg = self._dll.octa_last_error()
            raise OctaError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")

    def _call_profiled(self, name, *args):
        """OCTA_PROFILE=1: synchronise around every entry point and accumulate wall time per function
        (debug aid; serialises the stream)."""
        import time
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = getattr(self._dll, name)(*args)
        torch.cuda.synchronize()
        e = PROFILE.setdefault(name, [0.0, 0])
        e[0] += time.perf_counter() - t0
        e[1] += 1
        if rc != 0:
            msg = self._dll.octa_last_error()
            raise OctaError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")

    def __getattr__(self, name):
        # first use of an entry point: build its checked wrapper once and cache it on the instance (later lookups never get here;
        # the step makes ~1100 of these calls, so the wrapper is as thin as it can be)
        if name.startswith("octa_"):
            fn = getattr(self._dll, name)
            if self.signatures[name][0] is ctypes.c_int and name != "octa_version":
                if PROFILE is not None:
                    wrapper = lambda *a, _n=name: self.call(_n, *a)
                else:
                    err = self._dll.octa_last_error

                    def wrapper(*a, _fn=fn, _n=name):
                        rc = _fn(*a)
                        if rc != 0:
                            msg = err()
                            raise OctaError(f"{_n} failed ({rc}): {msg.decode() if msg else '?'}")
            else:
                wrapper = fn
            self.__dict__[name] = wrapper
            return wrapper
        raise AttributeError(name)


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        with _lock:
            if _LIB is None:
                _LIB = _Lib()
    return _LIB
