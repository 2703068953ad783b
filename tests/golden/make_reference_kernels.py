"""Lift the reference's own local-kernel TEXT (the C strings its PyOP2 tests hand to ``op2.Kernel``) into
tests/golden/reference_kernels.json, or verify the committed file against the reference.

The reference cannot be imported here (no loopy / petsc4py -- SURVEY.md 8c), but its test kernels are plain C strings that need
nothing but a C compiler: they are read out of the test sources with ``ast`` -- every ``op2.Kernel(<text>, "<name>")`` call whose
text resolves statically (a literal, a local variable bound to a literal, ``%`` / ``.format`` / f-strings over the scalar type
names) -- and stored UNCHANGED.  tests/test_reference_kernels.py / tests/test_gpu_reference_kernels.py feed them through the
oracle and through the HIP wrappers and assert the reference's own expected values.

    python tests/golden/make_reference_kernels.py          # verify (needs /root/reference)
    python tests/golden/make_reference_kernels.py --write  # regenerate
"""
import ast
import json
import os
import sys

REF = "/root/reference"
FILES = ["tests/pyop2/test_matrices.py", "tests/pyop2/test_indirect_loop.py", "tests/pyop2/test_extrusion.py",
         "tests/pyop2/test_subset.py", "tests/pyop2/test_direct_loop.py", "tests/pyop2/test_global_reduction.py",
         "tests/pyop2/test_iteration_space_dats.py", "tests/pyop2/test_vector_map.py"]
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_kernels.json")

# what the reference's datatype helpers evaluate to in its default configuration (pyop2/datatypes.py:6-10, 34-60:
# ScalarType = float64, IntType = int32; as_cstr maps them to "double" / "int32_t")
ENV = {"ScalarType_c": "double", "IntType_c": "int32_t", "valuetype": "double", "ScalarType": "double"}


class _Str(str):
    """ScalarType / IntType stand-ins: format like the C name"""


def _module_consts(tree):
    env = dict(ENV)
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            try:
                env[node.targets[0].id] = ast.literal_eval(node.value)
            except (ValueError, SyntaxError):
                pass
    return env


def _resolve(node, scope_assigns, env, depth=0):
    """the string an expression evaluates to, or None"""
    if depth > 6:
        return None
    if isinstance(node, ast.Constant) and isinstance(node.value, str):
        return node.value
    if isinstance(node, ast.Name):
        if node.id in scope_assigns:
            for value in reversed(scope_assigns[node.id]):
                got = _resolve(value, scope_assigns, env, depth + 1)
                if got is not None:
                    return got
        v = env.get(node.id)
        return v if isinstance(v, str) else None
    try:
        # % / .format / f-strings over names the environment knows
        names = {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}
        local = dict(env)
        for n in names:
            if n in scope_assigns:
                got = _resolve(ast.Name(id=n), scope_assigns, env, depth + 1)
                if got is not None:
                    local[n] = got
        val = eval(compile(ast.Expression(body=node), "<kernel>", "eval"), {"__builtins__": {}}, local)
        return val if isinstance(val, str) else None
    except Exception:
        return None


def extract(path):
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    env = _module_consts(tree)
    out = {}
    unresolved = []

    def visit_scope(fn, qual):
        assigns = {}
        for node in ast.walk(fn):
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                assigns.setdefault(node.targets[0].id, []).append(node.value)
        for node in ast.walk(fn):
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "Kernel"
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "op2" and len(node.args) >= 2):
                name = _resolve(node.args[1], assigns, env)
                code = _resolve(node.args[0], assigns, env)
                if name is None or code is None:
                    unresolved.append(f"{path}:{node.lineno} ({qual})")
                    continue
                key = f"{os.path.basename(path)}::{qual}::{name}"
                n, k = 1, key
                while k in out and out[k]["code"] != code:
                    n += 1
                    k = f"{key}#{n}"
                out[k] = {"name": name, "code": code, "source": f"{path}:{node.lineno}"}

    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            visit_scope(node, node.name)
        elif isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef):
                    visit_scope(sub, f"{node.name}.{sub.name}")
    return out, unresolved


def main():
    got, unresolved = {}, []
    for f in FILES:
        if not os.path.exists(os.path.join(REF, f)):
            continue
        o, u = extract(f)
        got.update(o)
        unresolved += u
    if "--write" in sys.argv:
        json.dump({"kernels": got, "not_static": unresolved}, open(OUT, "w"), indent=1, sort_keys=True)
        print(f"wrote {len(got)} kernels ({len(unresolved)} call sites whose text is not static) to {OUT}")
        return 0
    committed = json.load(open(OUT))["kernels"]
    bad = sorted(k for k in set(got) | set(committed) if got.get(k) != committed.get(k))
    if bad:
        print("MISMATCH vs the reference test sources:", bad)
        return 1
    print(f"reference_kernels.json matches the reference test sources: {len(got)} kernels")
    return 0


if __name__ == "__main__":
    sys.exit(main())
