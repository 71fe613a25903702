"""CPU oracle for the SynTalker denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy / CPU torch, the arithmetic of the reference's
``diffusion/*`` and ``models/denoiser*.py`` for the one path this repo accelerates.  It exists so
that the HIP path can be checked on a machine where ``/root/reference`` does not exist.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * nothing under ``syntalker_amd/`` imports it; the product path raises if the HIP library is
    missing rather than falling back to this code.

Pinning: the reference has no tests or golden vectors of its own.  The oracle is
pinned against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py`` (committed, with the vectors it produced under ``tests/golden/``);
``tests/test_oracle_golden.py`` replays them.
"""
