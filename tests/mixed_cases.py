"""Shared builders for the mixed-space and periodic-extrusion tests (CPU host-sim and GPU parity)."""
import numpy as np

from firedrake_amd import op2
from helpers import structured_tri_mesh

rdata = lambda s: np.arange(1, s + 1, dtype=np.float64)       # tests/pyop2/test_matrices.py:508

ADDONE_MAT = """static void addone_mat(PetscScalar v[9], double d[3]) {
            for (int i = 0; i < 3; i++)
               for (int j = 0; j < 3; j++)
                  v[i*3 + j] += d[i]*d[j];
        }"""                                                  # test_matrices.py:877-881
ADDONE_RHS = """
static void addone_rhs(PetscScalar v[3], double d[3]) {
  for (int i=0; i<3; ++i)
    v[i] += d[i];
}
        """                                                   # test_matrices.py:892-897
ADDONE_RHS_VEC = """
static void addone_rhs_vec(PetscScalar v[6], double d[6]) {
  for (int i=0; i<3; ++i) {
    v[i*2+0] += d[i*2+0];
    v[i*2+1] += d[i*2+1];
  }
}
        """                                                   # test_matrices.py:921-928

# golden blocks of TestMixedMatrices (test_matrices.py:863-871)
OD = np.array([[1.0, 2.0, 0.0, 0.0], [0.0, 4.0, 6.0, 0.0], [0.0, 0.0, 9.0, 12.0]])
LL = np.diag([1.0, 8.0, 18.0, 16.0]) + np.diag([2.0, 6.0, 12.0], -1) + np.diag([2.0, 6.0, 12.0], 1)


def reference_mixed_fixture():
    """mset, mdat, mvdat, mmap, msparsity of tests/pyop2/test_matrices.py:503-541."""
    mset = op2.MixedSet((op2.Set(3), op2.Set(4)))
    mdat = op2.MixedDat(op2.Dat(s, rdata(s.size)) for s in mset)
    mvdat = op2.MixedDat(op2.Dat(s ** 2, list(zip(rdata(s.size), rdata(s.size)))) for s in mset)
    elem, node = mset
    mmap = op2.MixedMap((op2.Map(elem, elem, 1, [0, 1, 2]), op2.Map(elem, node, 2, [0, 1, 1, 2, 2, 3])))
    msparsity = op2.Sparsity((mset ** 1, mset ** 1),
                             {(i, j): [(rm, cm, None)] for i, rm in enumerate(mmap) for j, cm in enumerate(mmap)})
    return mset, mdat, mvdat, mmap, msparsity


def velocity_pressure_space(nx, ny, vdim, seed=0):
    """A (P1^vdim x P0) pair on a triangle mesh: vertices carry ``vdim`` components, cells one value.
    Returns (cells set, MixedSet, MixedMap, coords Dat, vertex map)."""
    coords, cells = structured_tri_mesh(nx, ny, seed=seed, perturb=0.2)
    verts, ele = op2.Set(len(coords)), op2.Set(len(cells))
    vmap = op2.Map(ele, verts, 3, cells)
    cmap = op2.Map(ele, ele, 1, np.arange(len(cells)))
    mset = op2.MixedSet((verts, ele))
    return ele, mset, op2.MixedMap((vmap, cmap)), op2.Dat(verts ** 2, coords), vmap


def mixed_kernels(vdim):
    """Element kernels on the (3*vdim + 1)-dof mixed element: a dense matrix depending on the geometry and a
    residual depending on a mixed coefficient.  Every entry is distinct, so a misplaced block shows up."""
    n = 3 * vdim + 1
    jac = ("static void mixed_jac_%d(double *A, const double *x) {\n"
           "  double area = 0.5*fabs((x[2]-x[0])*(x[5]-x[1]) - (x[4]-x[0])*(x[3]-x[1]));\n"
           "  for (int i = 0; i < %d; ++i) for (int j = 0; j < %d; ++j)\n"
           "    A[i*%d + j] += area*(1.0 + 0.25*i + 0.125*j*j) + (i == j ? 1.0 : 0.0) + 0.01*x[(i + 2*j) %% 6];\n"
           "}\n" % (vdim, n, n, n))
    res = ("static void mixed_res_%d(double *b, const double *x, const double *w) {\n"
           "  double area = 0.5*fabs((x[2]-x[0])*(x[5]-x[1]) - (x[4]-x[0])*(x[3]-x[1]));\n"
           "  for (int i = 0; i < %d; ++i) { double s = 0.0; for (int j = 0; j < %d; ++j) s += (1.0 + 0.1*((i*7 + j*3) %% 5))*w[j];\n"
           "    b[i] += area*s + 0.5*w[i]; }\n"
           "}\n" % (vdim, n, n))
    return op2.Kernel(jac, "mixed_jac_%d" % vdim), op2.Kernel(res, "mixed_res_%d" % vdim)


def periodic_column_mesh(rng, nbase=5, ncl=4, nv=7):
    """Periodic columns of ``ncl`` cell layers and ``ncl`` node levels (the top level IS the bottom level): 3 base
    vertices x {lower, upper} per cell; the upper vertices of the top cell wrap to level 0, which is what
    offset_quotient = 1 on the upper entries expresses (pyop2/types/map.py:46-53, codegen/builder.py:108-120)."""
    base = op2.Set(nbase)
    ext = op2.ExtrudedSet(base, layers=ncl + 1, extruded_periodic=True)
    nodes = op2.Set(nv * ncl)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    vals = np.concatenate([tri * ncl, tri * ncl + 1], axis=1).astype(np.int32)
    cm = op2.Map(ext, nodes, 6, vals, offset=[1] * 6, offset_quotient=[0, 0, 0, 1, 1, 1])
    return base, ext, nodes, cm


def vector_p1_elasticity_kernel(dim=3):
    """Element matrix of  int eps(u):eps(v) + div(u) div(v) dx  on vector P1 simplices, laid out [i][p][j][q] like the MatPack
    of a VectorFunctionSpace (pyop2/codegen/builder.py:538-548): rows (node i, component p), columns (node j, component q)."""
    from firedrake_amd import op2
    from firedrake_amd.forms import _GEOM
    nv = dim + 1
    fact = 2 if dim == 2 else 6
    body = f"""
static void vec_elasticity(double *restrict A, const double *restrict x)
{{
{_GEOM[dim]}
  double g[{nv}][{dim}];
  for (int a = 0; a < {dim}; ++a) {{
    double s = 0.0;
    for (int i = 1; i < {nv}; ++i) {{ g[i][a] = K[i-1][a]; s -= K[i-1][a]; }}
    g[0][a] = s;
  }}
  const double vol = adet / {fact}.0;
  for (int i = 0; i < {nv}; ++i)
    for (int p = 0; p < {dim}; ++p)
      for (int j = 0; j < {nv}; ++j)
        for (int q = 0; q < {dim}; ++q) {{
          double gg = 0.0;
          for (int a = 0; a < {dim}; ++a) gg += g[i][a] * g[j][a];
          A[((i*{dim} + p)*{nv} + j)*{dim} + q] += vol * (0.5 * ((p == q) ? gg : 0.0) + 0.5 * g[i][q] * g[j][p] + g[i][p] * g[j][q]);
        }}
}}
"""
    return op2.Kernel(body, "vec_elasticity")


def q1_hex_helmholtz_kernel():
    """a(u, v) = int grad(u).grad(v) + u v dx on a trilinear hexahedron, Q1 basis, 2x2x2 Gauss points.  Arguments: A[8*8],
    coords[8*3]; vertex index a*4 + b*2 + c like firedrake_amd.mesh.make_extruded_hex_mesh (c along the extrusion)."""
    from firedrake_amd import op2
    body = """
static void q1_helmholtz(double *restrict A, const double *restrict x)
{
  static const double QP[2] = {0.21132486540518713, 0.7886751345948129};
  for (int q1 = 0; q1 < 2; ++q1) for (int q2 = 0; q2 < 2; ++q2) for (int q3 = 0; q3 < 2; ++q3) {
    const double t[3] = {QP[q1], QP[q2], QP[q3]};
    double N[8], dN[8][3], J[3][3] = {{0,0,0},{0,0,0},{0,0,0}};
    for (int v = 0; v < 8; ++v) {
      const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
      const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
      const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
      N[v] = Na*Nb*Nc; dN[v][0] = da*Nb*Nc; dN[v][1] = Na*db*Nc; dN[v][2] = Na*Nb*dc;
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) J[r][s] += x[3*v + r] * dN[v][s];
    }
    const double c00 = J[1][1]*J[2][2] - J[1][2]*J[2][1], c01 = J[1][2]*J[2][0] - J[1][0]*J[2][2], c02 = J[1][0]*J[2][1] - J[1][1]*J[2][0];
    const double det = J[0][0]*c00 + J[0][1]*c01 + J[0][2]*c02, id = 1.0 / det;
    const double K[3][3] = {
      { c00*id, (J[0][2]*J[2][1] - J[0][1]*J[2][2])*id, (J[0][1]*J[1][2] - J[0][2]*J[1][1])*id },
      { c01*id, (J[0][0]*J[2][2] - J[0][2]*J[2][0])*id, (J[0][2]*J[1][0] - J[0][0]*J[1][2])*id },
      { c02*id, (J[0][1]*J[2][0] - J[0][0]*J[2][1])*id, (J[0][0]*J[1][1] - J[0][1]*J[1][0])*id } };
    const double w = 0.125 * fabs(det);
    double g[8][3];
    for (int v = 0; v < 8; ++v) for (int r = 0; r < 3; ++r) {
      double s = 0.0;
      for (int a = 0; a < 3; ++a) s += dN[v][a] * K[a][r];
      g[v][r] = s;
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
      A[i*8 + j] += w * (g[i][0]*g[j][0] + g[i][1]*g[j][1] + g[i][2]*g[j][2] + N[i]*N[j]);
  }
}
"""
    return op2.Kernel(body, "q1_helmholtz")
