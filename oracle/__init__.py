"""CPU oracle for the OCTAve segmentor+discriminator training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product path (``octave_amd``) never imports this package and raises when
its HIP library is missing.

Parity pin: ``tests/golden/*.npz`` were produced by ``oracle/gen_golden.py``
by importing the reference from ``/root/reference`` on CPU (with the two
missing third-party imports of ``segmentor/losses.py`` -- ``kornia`` resize and
``loguru`` -- replaced in memory, see that script).  ``tests/test_oracle.py``
checks every function here against those vectors.  The single call that is
NOT pinned by the reference is kornia's nearest ``resize`` (kornia is absent
and the reference has no tests): "parity unpinned" for that one call, which
for the integer ratios of the hot path is ``src = dst // f``.
"""
