import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firedrake_amd import forms, mesh as fmesh
from firedrake_amd.configuration import configuration
for nb in ("lexicographic", "random"):
    mesh = fmesh.UnitCubeMesh((215,)*3, degrees=(1,), perturb=0.1, tile=(8, 8, 4), numbering=nb)
    for cap in (3840, 4224, 4416, 4608):
        configuration["ocr_nnz_per_block_ordered"] = cap
        prob = forms.PoissonProblem(mesh, 1, bcs=True)
        prob.assemble_jacobian()
        loop = prob.jacobian()[1]
        geos = [g for key, g in loop._prepared["parts"].items() if key[0] == "ocr" and isinstance(g, dict)]
        g = geos[0]
        op = g.get("op") or g.get("plan")
        print(nb, cap, "lds", g.get("lds"), "blocks", getattr(op, "nblocks", None), "max_nnz", getattr(op, "max_nnz", None), "ninst", getattr(op, "ninst", None), {k: type(v).__name__ for k, v in g.items()} if cap == 3840 and nb == "lexicographic" else "", flush=True)
        del prob
