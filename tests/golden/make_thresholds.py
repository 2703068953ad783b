"""Regenerate / verify tests/golden/reference_thresholds.json: the pass criteria the reference's own regression tests
assert for the variational problems tests/test_reference_thresholds.py reproduces.  Lifted from the reference's test
sources with ``ast`` (Firedrake itself cannot be imported here), never retyped.

    python tests/golden/make_thresholds.py          # verify against the committed JSON
    python tests/golden/make_thresholds.py --write
"""
import ast
import json
import os
import sys

REF = "/root/reference/tests/firedrake"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_thresholds.json")


def _func(tree, name):
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return node
    raise KeyError(name)


def _parametrize(fn, argname_fragment):
    """literal value list of the @pytest.mark.parametrize decorator whose argnames mention ``argname_fragment``"""
    for d in fn.decorator_list:
        if isinstance(d, ast.Call) and getattr(d.func, "attr", "") == "parametrize":
            names = ast.literal_eval(d.args[0])
            if argname_fragment in (names if isinstance(names, str) else ",".join(names)):
                return ast.literal_eval(d.args[1])
    raise KeyError(argname_fragment)


def _compare_constants(fn):
    """numeric constants that appear as the right-hand comparator of a comparison inside ``fn``, with the operator"""
    out = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Compare) and isinstance(node.comparators[0], ast.Constant) and isinstance(node.comparators[0].value, (int, float)):
            out.append((type(node.ops[0]).__name__, float(node.comparators[0].value)))
    return out


def main():
    got = {}
    # ---- test_helmholtz.py: triangles, CG2, UnitSquareMesh(2^r) for r in range(3, 6), orders > 2.8
    path = os.path.join(REF, "regression/test_helmholtz.py")
    tree = ast.parse(open(path).read())
    fn = _func(tree, "run_firedrake_helmholtz")
    rng = [ast.literal_eval(a) for n in ast.walk(fn) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "range" for a in n.args]
    (op, thr), = _compare_constants(fn)
    degree_default = ast.literal_eval(_func(tree, "helmholtz").args.defaults[1])
    got["helmholtz_triangles"] = {"source": "tests/firedrake/regression/test_helmholtz.py:56-62", "degree": degree_default,
                                  "refinements": list(range(*rng)), "op": op, "min_order": thr}
    fq = _func(tree, "test_firedrake_helmholtz_scalar_convergence_on_quadrilaterals")
    got["helmholtz_quadrilaterals"] = {"source": "tests/firedrake/regression/test_helmholtz.py:73-83",
                                       "cases": [{"degree": c[0][0], "refinements": list(range(*c[0][1])), "min_order": c[1]}
                                                 for c in _parametrize(fq, "convrate")]}
    # ---- extrusion/test_helmholtz_scalar.py: CG1..3 on extruded meshes
    path = os.path.join(REF, "extrusion/test_helmholtz_scalar.py")
    fe = _func(ast.parse(open(path).read()), "test_scalar_convergence")
    got["helmholtz_extruded"] = {"source": "tests/firedrake/extrusion/test_helmholtz_scalar.py:8-33",
                                 "cases": [{"family": c[0][0], "degree": c[0][1], "refinements": list(range(*c[0][2])), "min_order": c[1]}
                                           for c in _parametrize(fe, "convrate")]}
    # ---- test_poisson_strong_bcs.py: error bounds of the nonlinear and the linear variant, degrees, refinement
    path = os.path.join(REF, "regression/test_poisson_strong_bcs.py")
    tree = ast.parse(open(path).read())
    out = {"source": "tests/firedrake/regression/test_poisson_strong_bcs.py:67-86"}
    for key, name in (("nonlinear", "test_poisson_analytic"), ("linear", "test_poisson_analytic_linear")):
        fn = _func(tree, name)
        (op, thr), = _compare_constants(fn)
        call = next(n for n in ast.walk(fn) if isinstance(n, ast.Call) and getattr(n.func, "id", "").startswith("run_test"))
        degs = None
        for d in fn.decorator_list:
            for n in ast.walk(d):
                if isinstance(n, ast.comprehension) and getattr(n.target, "id", "") == "d":
                    degs = list(ast.literal_eval(n.iter))
        out[key] = {"op": op, "max_error": thr, "refinement": ast.literal_eval(call.args[0]), "degrees": degs}
    got["poisson_strong_bcs"] = out
    if "--write" in sys.argv:
        json.dump(got, open(OUT, "w"), indent=1)
        print("rewritten", OUT)
        return 0
    committed = json.load(open(OUT))
    bad = [k for k in got if committed.get(k) != got[k]]
    if bad:
        print("MISMATCH:", bad)
        return 1
    print("reference thresholds verified:", ", ".join(sorted(got)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
