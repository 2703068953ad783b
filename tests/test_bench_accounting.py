"""CPU: the byte accounting behind ``roofline.achieved`` in bench.py is the one SURVEY.md 8(d) defines (every input array
read once, every output written once, plus the zeroing pass) -- pinned to the figures quoted there for config C2 --
and the C2 mesh sizes are the ones the scope table states."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_c2_algorithmic_bytes_match_the_scope_table():
    b = _bench()
    N = 215
    ncell, nnode = 6 * N ** 3, (N + 1) ** 3
    nnz = nnode + 2 * (7 * N ** 3 + 9 * N ** 2 + 3 * N)          # vertices + 2 * edges
    assert (ncell, nnode, nnz) == (59630250, 10077696, 150048286)
    # residual with ONE coefficient: map 954 MB + coords 242 MB + u 81 MB + output 81 MB + zeroing 81 MB = 1.44 GB
    r1 = b.algorithmic_bytes(ncell, 4, nnode, 3, 1)
    assert abs(r1 - 1.44e9) < 0.01e9
    assert r1 == ncell * 16 + nnode * 24 + nnode * 8 + 2 * nnode * 8
    # the benchmark's residual reads two coefficients (u and f): + 81 MB
    assert b.algorithmic_bytes(ncell, 4, nnode, 3, 2) == r1 + nnode * 8 == 1518434976
    # Jacobian: map + coords + values + zeroing = 954 MB + 242 MB + 1.20 GB + 1.20 GB = 3.60 GB
    j = b.algorithmic_bytes(ncell, 4, nnode, 3, 0, nnz)
    assert j == 3596721280 and abs(j - 3.60e9) < 0.01e9
    assert b.HBM_PEAK_GBS == 8000.0
