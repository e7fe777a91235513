"""Closed-form weight / input fill used by every fixture (test infrastructure).

The implementation lives in ``octave_amd/synth.py`` (pure numpy/torch data generation, no oracle code): ``bench.py``'s
``dice_vs_ref`` leg needs the same closed-form weights to evaluate the HIP path on a reference fixture's input, and nothing
outside ``tests/``, ``smoke()`` and the ``cpu_baseline`` leg may import ``oracle/``.  This module keeps the historical import
path of the tests and of ``oracle/gen_golden.py``; the fixtures are bit-identical (same functions)."""
from octave_amd.synth import COND_SCALE, _hash_uniform, fill_state_dict, fill_tensor, hash_input  # noqa: F401
