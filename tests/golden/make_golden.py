"""Regenerate / verify tests/golden/pyop2_matrices.json from the reference's test file.

Runs only where /root/reference exists (the build container).  The reference
(pyop2) cannot be *imported* here (no loopy/petsc4py -- SURVEY.md 8c), so the
golden numbers are lifted from the literals in the reference's own test source
with ``ast`` and checked against the committed JSON.

    python tests/golden/make_golden.py          # verify
    python tests/golden/make_golden.py --write  # rewrite the expected_* entries
"""
import ast
import json
import os
import re
import sys

REF = "/root/reference/tests/pyop2/test_matrices.py"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pyop2_matrices.json")


def literal_in_fixture(tree, name):
    """First list/tuple literal inside fixture function ``name``."""
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            for sub in ast.walk(node):
                if isinstance(sub, (ast.List, ast.Tuple)):
                    try:
                        return ast.literal_eval(sub)
                    except ValueError:
                        continue
    raise KeyError(name)


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    got = {
        "expected_matrix": [list(r) for r in literal_in_fixture(tree, "expected_matrix")],
        "expected_vector_matrix": [list(r) for r in literal_in_fixture(tree, "expected_vector_matrix")],
        "expected_rhs": list(literal_in_fixture(tree, "expected_rhs")),
        "expected_vec_rhs": [list(r) for r in literal_in_fixture(tree, "expected_vec_rhs")],
        "coords": [list(r) for r in literal_in_fixture(tree, "coords")],
        "f": list(literal_in_fixture(tree, "f")),
    }
    m = re.search(r"elem_node_map = np.asarray\(\[([0-9, ]+)\]", src)
    flat = [int(x) for x in m.group(1).split(",")]
    got["elem_node_map"] = [flat[0:3], flat[3:6]]
    w = re.search(r"double w\[6\] = \{([^}]*)\}", src).group(1)
    got["quad6_weights"] = [float(x) for x in w.replace("\n", " ").split(",")]
    cg = re.search(r"double CG1\[3\]\[6\] = \{(.*?)\} \};", src, re.S).group(1)
    vals = [float(x) for x in re.findall(r"[-+]?\d*\.\d+", cg)]
    got["quad6_basis"] = [vals[0:6], vals[6:12], vals[12:18]]
    committed = json.load(open(OUT))
    if "--write" in sys.argv:
        committed.update(got)
        json.dump(committed, open(OUT, "w"), indent=1)
        print("rewritten", OUT)
        return 0
    bad = [k for k, v in got.items() if committed[k] != v]
    if bad:
        print("MISMATCH vs reference literals:", bad)
        return 1
    print("golden JSON matches the reference test literals:", sorted(got))
    return 0


if __name__ == "__main__":
    sys.exit(main())
