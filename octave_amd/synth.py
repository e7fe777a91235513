"""Deterministic closed-form parameter / input fill (synthetic data, no arithmetic of the path).

The real weights are 292 MB and cannot be committed, so golden whole-network
vectors use a fill that is a pure function of (state_dict key, element index):
an integer hash mapped to a uniform value with the variance the reference's
own init would give the tensor (extra/resnest.py:368-374 for conv/BN).  Pure
uint64 numpy arithmetic, so it is bit-identical on every machine.
"""
import zlib
import numpy as np
import torch


def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1) from a splitmix-style integer hash."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (i + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    u = (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)  # [0,1)
    return 2.0 * u - 1.0


def fill_tensor(name: str, t: torch.Tensor, salt: int = 0) -> torch.Tensor:
    """Value for state_dict entry `name` with the shape/dtype of `t`."""
    seed = (zlib.crc32(name.encode()) + 1000003 * salt) & 0xFFFFFFFF
    n = t.numel()
    if not t.is_floating_point():           # num_batches_tracked
        return torch.zeros_like(t)
    u = _hash_uniform(n, seed).reshape(tuple(t.shape))
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_mean":
        v = 0.05 * u
    elif leaf == "running_var":
        v = 1.0 + 0.1 * u
    elif leaf in ("weight_u", "weight_v"):   # spectral-norm vectors: unit norm
        v = u / np.sqrt((u * u).sum() + 1e-12)
    elif t.dim() == 1 and leaf == "weight":  # BN gamma
        v = 1.0 + 0.1 * u
    elif t.dim() == 1:                       # biases / BN beta
        v = 0.05 * u
    elif t.dim() == 4:                       # conv: var = 2 / (k*k*Cout) like the reference init
        fan = t.shape[2] * t.shape[3] * t.shape[0]
        v = u * np.sqrt(3.0 * 2.0 / fan)
    else:                                    # linear
        v = u * np.sqrt(3.0 / t.shape[-1])
    return torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)


@torch.no_grad()
def fill_state_dict(sd, salt: int = 0):
    """In-place deterministic fill of every entry of a state_dict-like mapping."""
    for k, v in sd.items():
        v.copy_(fill_tensor(k, v, salt))
    return sd


def hash_input(shape, seed: int, lo=0.0, hi=1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    u = 0.5 * (_hash_uniform(n, seed) + 1.0)
    return torch.from_numpy((lo + (hi - lo) * u).reshape(shape)).float()
