"""oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (PyOP2 generated wrapper + PETSc
insertion + sparsity + halo protocol).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline -- never as the
thing measured or shipped.  ``firedrake_amd`` must not import it.

Parity status: the *wrapper* semantics are pinned against the reference's own
golden vectors (tests/pyop2/test_matrices.py:463-501 etc., see
tests/test_oracle_golden.py).  The hand-restated TSFC-equivalent *local
kernels* for the five configs are "parity unpinned" at the element-tensor level
(the reference stores no element tensors and TSFC/FIAT cannot be imported
here -- SURVEY.md 8c); they are validated by the analytic identities the
reference's regression tests use.
"""
from .wrapper import (ODat, OGlobal, OMat, OMixedDat, OMixedMat, OracleCSR, par_loop, build_sparsity, make_owner_partition,  # noqa: F401
                      READ, WRITE, RW, INC, MIN, MAX, ALL, ON_BOTTOM, ON_TOP,
                      ON_INTERIOR_FACETS, generate_wrapper, compile_c)
