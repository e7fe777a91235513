"""Layout constants shared by oracle/gen_golden.py and the tests that read its fixtures (test infrastructure)."""


def grad_stride(numel: int, cap: int = 40000) -> int:
    """Full gradients beyond `cap` elements are stored as flat[::stride] of the OIHW-logical order (trainstep_400c.npz)."""
    return max(1, -(-numel // cap))
