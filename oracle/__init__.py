"""CPU oracle for the ADMP multipolar-PME hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``admp_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Pinning status
--------------
* geometry helpers (pbc shift, local frames, quasi-internal frame, harmonic
  rotations, cart->harm conversion): PINNED by the reference's own unit-test
  vectors (reference ``tests/test_sptial.py``, ``tests/test_multipole.py``;
  the literal arrays are committed as data in ``tests/golden/ref_unit_vectors.json``).
* hot path proper (``pme.py`` / ``recip.py`` / ``disp_pme.py`` / ``pairwise.py``
  energies and gradients): **PARITY UNPINNED** -- the reference holds no test
  or valid golden for these (its ``examples/*/ref_out`` are stale with respect
  to the shipped inputs) and it cannot be executed here
  (needs ``jax``/``jax_md``: ``ModuleNotFoundError``, no wheel, no network).
  The restatement is instead validated by: Ewald-kappa independence, an
  independent Cartesian-tensor direct sum on isolated clusters, central finite
  differences of every gradient, and agreement with the values the survey
  derived independently.
"""
