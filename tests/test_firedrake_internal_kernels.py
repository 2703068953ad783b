"""Firedrake's OWN C-string kernels -- the few parloops Firedrake writes by hand instead of generating -- lifted verbatim and run
through the backend's wrappers against the oracle:

* ``multiplicity`` (firedrake/interpolation.py:1066-1071, firedrake/preconditioners/pmg.py:1149-1154): ``w[i] += 1`` over the
  cell-node map, INC -- the DoF multiplicity whose reciprocal weights a Cofunction interpolation / a p-multigrid restriction;
* ``copy`` (firedrake/preconditioners/facet_split.py:259-268): an int32 Dat WRITTEN through the cell-node map of one space from an
  int32 Dat READ through a PermutedMap of another -- the facet-split permutation.

The texts are not ``static``, use ``PetscScalar`` / ``PetscInt`` and ``restrict`` as Firedrake wrote them.  CPU: host-sim of the
generated wrappers; GPU: the backend-picked wrapper and the direct one."""
import numpy as np
import pytest

from firedrake_amd import mesh as fmesh, op2
from firedrake_amd.configuration import configuration
from helpers import oracle_run

# firedrake/interpolation.py:1066-1069 == firedrake/preconditioners/pmg.py:1149-1152 (f-string with wsize substituted)
MULTIPLICITY = """
        void multiplicity(PetscScalar *restrict w) {{
            for (PetscInt i=0; i<{wsize}; i++) w[i] += 1;
        }}"""
# firedrake/preconditioners/facet_split.py:259-262
COPY = """
    void copy(PetscInt *restrict w, const PetscInt *restrict v) {{
        for (PetscInt i=0; i<{wsize}; i++) w[i] = v[i];
    }}"""


def _spaces():
    m = fmesh.UnitCubeMesh(5, degrees=(1, 2), perturb=0.1, numbering="random")
    return m, m.space(2)


def _multiplicity_loop(m, V, block=1):
    w = op2.Dat(V.node_set ** block if block > 1 else V.node_set)
    k = op2.Kernel(MULTIPLICITY.format(wsize=V.cell_node_map.arity * block), "multiplicity")
    return w, k, (w(op2.INC, V.cell_node_map),)


def _copy_loop(m, V):
    """W = V with its element DoFs permuted (what restricted_dofs gives for a facet split): v = arange read through a PermutedMap,
    written through the plain cell-node map -- the result is the permutation of the global numbering"""
    nn = V.node_set.total_size
    vdat = op2.Dat(V.node_set, np.arange(nn, dtype=np.int32), np.int32)
    wdat = op2.Dat(V.node_set, np.full(nn, -1, dtype=np.int32), np.int32)
    eperm = np.array([4, 5, 6, 7, 8, 9, 0, 1, 2, 3])                    # edge DoFs first, then the vertices
    pmap = op2.PermutedMap(V.cell_node_map, eperm)
    k = op2.Kernel(COPY.format(wsize=V.cell_node_map.arity), "copy", requires_zeroed_output_arguments=False)
    return wdat, k, (wdat(op2.WRITE, V.cell_node_map), vdat(op2.READ, pmap))


def test_firedrake_internal_kernels_on_host():
    from firedrake_amd.codegen import select_mode
    from hostsim import run_direct, run_staged
    m, V = _spaces()
    for block in (1, 3):
        w, k, args = _multiplicity_loop(m, V, block)
        pl = op2.LegacyParloop(k, m.cell_set, *args)
        ref = oracle_run(k, m.cell_set, *args)[0]
        assert ref.min() >= 1 and ref.max() > 4                         # every DoF belongs to a cell, vertices to many
        assert select_mode(pl.global_kernel) == "staged"
        assert np.array_equal(run_staged(pl, epb=200)[0], ref) and np.array_equal(run_direct(pl)[0], ref)
    wdat, k, args = _copy_loop(m, V)
    pl = op2.LegacyParloop(k, m.cell_set, *args)
    ref = oracle_run(k, m.cell_set, *args)[0]
    assert select_mode(pl.global_kernel) == "staged"                    # READ through the PermutedMap staged, WRITE from the lane
    got = run_staged(pl, epb=150)[0]
    # every node is written by every cell that holds it -- with the value of ANOTHER local DoF of that cell, so the winner is
    # whichever cell came last (pyop2 WRITE semantics): compare the set of admissible values instead of one sequential order
    cm = np.asarray(V.cell_node_map.values_with_halo)
    allowed = [set() for _ in range(V.node_set.total_size)]
    for row in cm:
        for i, n in enumerate(row):
            allowed[n].add(int(row[[4, 5, 6, 7, 8, 9, 0, 1, 2, 3][i]]))
    assert all(int(g) in a for g, a in zip(got, allowed)) and all(int(r) in a for r, a in zip(ref, allowed))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "direct"])
def test_firedrake_internal_kernels_on_the_gpu(mode, monkeypatch):
    monkeypatch.setitem(configuration, "mode", mode)
    m = fmesh.UnitCubeMesh(24, degrees=(1, 2), perturb=0.1, numbering="lexicographic")
    V = m.space(2)
    for block in (1, 3):
        w, k, args = _multiplicity_loop(m, V, block)
        ref = oracle_run(k, m.cell_set, *args)[0]
        pl = op2.LegacyParloop(k, m.cell_set, *args)
        pl()
        assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
        assert np.array_equal(np.asarray(w.data_ro).reshape(ref.shape), ref)
    wdat, k, args = _copy_loop(m, V)
    pl = op2.LegacyParloop(k, m.cell_set, *args)
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    got = np.asarray(wdat.data_ro)
    cm = np.asarray(V.cell_node_map.values_with_halo)
    perm = np.array([4, 5, 6, 7, 8, 9, 0, 1, 2, 3])
    # admissible: the value some cell holding the node writes there
    ok = np.zeros(len(got), dtype=bool)
    for i in range(10):
        np.logical_or.at(ok, cm[:, i], got[cm[:, i]] == cm[:, perm[i]])        # (.at: a node appears in many rows)
    assert ok.all()
