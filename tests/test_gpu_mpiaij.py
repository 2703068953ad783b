"""GPU: the hand-over of an assembled matrix to a DISTRIBUTED PETSc matrix: the owned rows of the device CSR
split into the diagonal / off-diagonal blocks of an MPIAIJ matrix -- the arrays of MatCreateMPIAIJWithSplitArrays -- with the
preallocation counts ``Sparsity.nnz`` / ``onnz`` of the reference (pyop2/types/mat.py:254-278).  Checked against the oracle's
pattern and values split with numpy.  PETSc itself is not installed; everything up to the one PETSc call is executed."""
import numpy as np
import pytest

from firedrake_amd import forms, mesh as fmesh, op2
from helpers import oracle_pattern, oracle_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree,rank,nranks,partition", [(1, 0, 2, "slabs"), (1, 1, 3, "slabs"), (2, 3, 4, "blocks"), (1, 0, 1, "slabs")])
def test_split_of_the_owned_rows_against_the_oracle(degree, rank, nranks, partition):
    m = fmesh.UnitCubeMesh(4, degrees=(degree,), tile=(2, 2, 2), perturb=0.1, rank=rank, nranks=nranks, partition=partition)
    prob = forms.PoissonProblem(m, degree, bcs=True)
    V = prob.V
    nown, ntot = V.node_set.size, V.node_set.total_size
    # global numbering of the local columns: the lattice index of every node (what the column lgmap of dataset.py:120-164 holds)
    p = degree * 4 + 1
    ijk = np.rint(V.node_points * (p - 1)).astype(np.int64)
    col_global = ((ijk[:, 2] * p + ijk[:, 1]) * p + ijk[:, 0]).astype(np.int32)
    mat = prob.assemble_jacobian()
    sp = mat.sparsity
    split, dv, ov = mat.mpiaij_values(col_global)
    # oracle: pattern (sparsity.pyx restated) and values, split with numpy
    mp, loop = prob.jacobian()
    lg = loop.arguments[0].lgmaps
    cm, xm = V.cell_node_map, m.coord_space.cell_node_map
    ref = oracle_run(prob.kjac, _all_cells(m), mp(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm))[0]
    for row in prob.bc_nodes:
        k = ref.rowptr[row] + np.searchsorted(ref.colidx[ref.rowptr[row]:ref.rowptr[row + 1]], row)
        ref.values[k] = 1.0
    A = ref.toscipy().tocsr()[:nown]
    D, O = A[:, :nown].tocsr(), A[:, nown:].tocsr()
    # explicit zeros are part of the pattern: compare index arrays, not just values
    pat = oracle_pattern(sp)
    rp, ci = pat.rowptr[:nown + 1], pat.colidx
    d_cnt = np.array([(ci[rp[r]:rp[r + 1]] < nown).sum() for r in range(nown)])
    o_cnt = np.diff(rp) - d_cnt
    assert np.array_equal(sp.nnz, d_cnt) and np.array_equal(sp.onnz, o_cnt)
    assert split.d_nnz == d_cnt.sum() and split.o_nnz == o_cnt.sum() and (nranks > 1) == (split.o_nnz > 0)
    drp, orp = split.d_rowptr.download(np.int32, (nown + 1,)), split.o_rowptr.download(np.int32, (nown + 1,))
    dci, oci = split.d_colidx.download(np.int32, (max(split.d_nnz, 1),))[:split.d_nnz], split.o_colidx.download(np.int32, (max(split.o_nnz, 1),))[:split.o_nnz]
    assert np.array_equal(drp, np.concatenate([[0], np.cumsum(d_cnt)])) and np.array_equal(orp, np.concatenate([[0], np.cumsum(o_cnt)]))
    exp_d = np.concatenate([ci[rp[r]:rp[r] + d_cnt[r]] for r in range(nown)])
    # off-diagonal block: GLOBAL columns, every row SORTED (MatCreateSeqAIJWithArrays needs sorted rows; the local ghost numbering
    # is not monotone in the global one)
    perm = [np.argsort(col_global[ci[rp[r] + d_cnt[r]:rp[r + 1]]], kind="stable") for r in range(nown)]
    exp_o = np.concatenate([col_global[ci[rp[r] + d_cnt[r]:rp[r + 1]]][perm[r]] for r in range(nown)]) if split.o_nnz else np.zeros(0, np.int32)
    assert np.array_equal(dci, exp_d) and np.array_equal(oci, exp_o)                 # diagonal block: local columns
    assert all(np.all(np.diff(oci[orp[r]:orp[r + 1]]) > 0) for r in range(nown))
    dval = dv.download(np.float64, (max(split.d_nnz, 1),))[:split.d_nnz]
    oval = ov.download(np.float64, (max(split.o_nnz, 1),))[:split.o_nnz]
    vals = ref.values
    exp_dv = np.concatenate([vals[rp[r]:rp[r] + d_cnt[r]] for r in range(nown)])
    exp_ov = np.concatenate([vals[rp[r] + d_cnt[r]:rp[r + 1]][perm[r]] for r in range(nown)]) if split.o_nnz else np.zeros(0)
    tol = 1e-12 * np.abs(vals).max()
    assert np.abs(dval - exp_dv).max() <= tol and (split.o_nnz == 0 or np.abs(oval - exp_ov).max() <= tol)
    assert abs(D.sum() + O.sum() - (dval.sum() + oval.sum())) <= 1e-9 * max(1.0, abs(D.sum()))
    # values refreshed after a second assembly into the same pattern
    prob.assemble_jacobian()
    _, dv2, _ = mat.mpiaij_values(col_global)
    assert dv2.ptr == dv.ptr and np.abs(dv2.download(np.float64, (split.d_nnz,)) - exp_dv).max() <= tol


def _all_cells(m):
    """the iteration set of the owner-computes-rows Jacobian: owned + ghost cells (non-owned rows are masked by the lgmaps)"""
    return _Whole(m.cell_set)


class _Whole:
    """view of a Set whose ``size`` is its total size (oracle_run iterates [0, size))"""

    def __init__(self, s):
        self._s = s
        self.size = s.total_size

    def __getattr__(self, k):
        return getattr(self._s, k)
