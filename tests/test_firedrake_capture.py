"""Replay of Firedrake captures (tools/firedrake_capture.py -> tests/golden/firedrake_<cfg>.npz): every parloop ``assemble()`` executed
-- TSFC's kernel text, PyOP2's maps and data, Firedrake's BC lgmaps -- runs through this repository's wrappers and must reproduce what
the reference produced, at the SURVEY.md Appendix D tolerances.  CPU: the oracle against the capture (pins the oracle to the reference
itself once a real capture exists); GPU: the HIP wrappers against the capture.  Until an environment with Firedrake has produced one, the
synthetic capture of tests/golden/make_synthetic_capture.py (same format, this repository's own kernels) keeps the path exercised; real
captures are picked up by file name, nothing else changes.  Matches pyop2/local_kernel.py:210-227, pyop2/parloop.py:243-260."""
import glob
import os

import numpy as np
import pytest

import capture_replay
from firedrake_amd import op2
from helpers import oracle_run

HERE = os.path.dirname(os.path.abspath(__file__))
CAPTURES = sorted(glob.glob(os.path.join(HERE, "golden", "firedrake_*.npz")))


def _check(tol):
    def check(got, expected, what):
        n = min(len(got), len(expected))
        scale = max(1.0, float(np.abs(expected).max()) if len(expected) else 1.0)
        assert np.abs(got[:n] - expected[:n]).max() <= tol * scale, what
    return check


def _compare_outputs(meta, z, mats, vec_after):
    tol = meta["tolerances"]
    out = meta["outputs"]
    if "J" in out and mats.get("J") is not None:
        csr = mats["J"]
        ref = [np.asarray(z[out["J"][k]]) for k in ("indptr", "indices", "data")]
        # rows the row lgmap masks hold only what the BC post-processing puts there (assemble.py:1501-1507): compared on the other rows
        jl = [lp for lp in meta["parloops"] if lp["form"] == "J"][-1]
        lg = next((a["lgmaps"] for a in jl["args"] if a["kind"] == "mat"), None)
        rows = None
        if lg is not None:
            rows = np.nonzero(np.asarray(z[lg[0]])[:len(ref[0]) - 1] >= 0)[0]
        worst = capture_replay.csr_rows_equal(np.asarray(csr[0]), np.asarray(csr[1]), np.asarray(csr[2]), *ref, rows=rows)
        assert worst <= tol["matrix"] * max(1.0, np.abs(ref[2]).max()), worst
    if "F" in out and vec_after.get("F") is not None:
        ref = np.asarray(z[out["F"]["data"]]).reshape(-1)
        got = vec_after["F"][:len(ref)]
        live = np.ones(len(ref), dtype=bool)
        jl = [lp for lp in meta["parloops"] if lp["form"] == "J"]
        lg = next((a["lgmaps"] for lp in jl for a in lp["args"] if a["kind"] == "mat" and a.get("lgmaps")), None)
        if lg is not None and len(np.asarray(z[lg[0]])) >= len(ref):
            live = np.asarray(z[lg[0]])[:len(ref)] >= 0           # BC rows are zeroed after the loops (bcs.py:192-221)
        assert np.abs(got[live] - ref[live]).max() <= tol["vector"] * max(1.0, np.abs(ref).max())


def test_a_capture_exists_and_is_well_formed():
    assert CAPTURES, "tests/golden/make_synthetic_capture.py writes the synthetic capture"
    for path in CAPTURES:
        meta, z = capture_replay.load(path)
        assert meta["parloops"] and all(lp["kernel_c"] and lp["kernel_name"] in lp["kernel_c"] for lp in meta["parloops"])
        for lp in meta["parloops"]:
            for a in lp["args"]:
                for key in [a.get("before"), a.get("after")] + list(a.get("lgmaps") or []):
                    assert key is None or key in z.files


@pytest.mark.parametrize("path", CAPTURES, ids=[os.path.basename(p) for p in CAPTURES])
def test_oracle_reproduces_the_capture(path):
    """CPU: the oracle (CPU restatement of the PyOP2 wrapper) on the CAPTURED kernel text and data gives the captured outputs."""
    meta, z = capture_replay.load(path)
    if meta.get("full_size"):
        pytest.skip("full-size captures are replayed on the GPU only")
    vec_after = {}

    def runner(kernel, iterset, args, loop):
        region = None if loop["iteration_region"] in ("ALL", "None", None) else getattr(op2, loop["iteration_region"].split(".")[-1], None)
        outs = oracle_run(kernel, iterset, *args, iteration_region=region, pass_layer_arg=loop["pass_layer_arg"])
        res = []
        for a, o in zip(loop["args"], outs):
            if a["kind"] == "mat":
                res.append((o.rowptr, o.colidx, o.values))
            else:
                res.append(o)
                if a["access"] != "READ" and a["kind"] == "dat":
                    vec_after[loop["form"]] = np.asarray(o).reshape(-1)
        return res

    mats = capture_replay.Replay(meta, z).run(runner, check=_check(meta["tolerances"]["vector"]))
    _compare_outputs(meta, z, mats, vec_after)


@pytest.mark.gpu
@pytest.mark.parametrize("path", CAPTURES, ids=[os.path.basename(p) for p in CAPTURES])
def test_hip_wrappers_reproduce_the_capture(path):
    """GPU: the same replay through ``op2.par_loop`` -- the wrapper the backend picks for every captured loop."""
    meta, z = capture_replay.load(path)
    vec_after = {}

    def runner(kernel, iterset, args, loop):
        region = None if loop["iteration_region"] in ("ALL", "None", None) else getattr(op2, loop["iteration_region"].split(".")[-1], None)
        kw = {} if region is None else {"iteration_region": region}
        if loop["pass_layer_arg"]:
            kw["pass_layer_arg"] = True
        op2.par_loop(kernel, iterset, *args, **kw)
        res = []
        for a, arg in zip(loop["args"], args):
            if a["kind"] == "mat":
                res.append(arg.data.csr())
            elif a["access"] == "READ":
                res.append(None)
            else:
                d = arg.data
                o = np.array(d.data_ro_with_halos if hasattr(d, "data_ro_with_halos") else d.data_ro)
                res.append(o)
                if a["kind"] == "dat":
                    vec_after[loop["form"]] = o.reshape(-1)
        return res

    mats = capture_replay.Replay(meta, z).run(runner, check=_check(meta["tolerances"]["vector"]))
    _compare_outputs(meta, z, mats, vec_after)
