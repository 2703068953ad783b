"""A/B of the MFMA matrix template's panel shape (configuration tp_max_panel_tiles / tp_chunk_tiles: whole 16-row panels against column
chunks) on Q4 (n = 32, config C3), Q5 (n = 16) and Q6 (n = 12): kernel time and fraction of the fp64 MFMA peak.  One process per
setting (compiled wrappers are cached per kernel; the constants are not part of the key).
python tools/ab_tensor_panels.py            # the sweep
python tools/ab_tensor_panels.py degree n nq panel chunk"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if len(sys.argv) == 1:
    for degree, n, nq in ((4, 32, 5), (5, 16, 6), (6, 12, 8)):
        for rep in range(2):
            for panel, chunk in ((14, 8), (8, 8), (4, 4)) if degree < 6 else ((8, 8), (8, 6), (8, 11)):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(degree), str(n), str(nq), str(panel), str(chunk)],
                                   capture_output=True, text=True)
                print((r.stdout.strip().splitlines() or ["FAILED: " + r.stderr[-300:]])[-1], flush=True)
    sys.exit(0)

import bench                                                                # noqa: E402
from firedrake_amd import _lib, forms, mesh as fmesh                       # noqa: E402
from firedrake_amd.configuration import configuration                      # noqa: E402
from firedrake_amd.device import Event                                      # noqa: E402

degree, n, nq, panel, chunk = (int(v) for v in sys.argv[1:6])
configuration["tp_max_panel_tiles"], configuration["tp_chunk_tiles"] = panel, chunk
m = fmesh.make_extruded_hex_mesh(n, n, degree, perturb=0.1)
prob = forms.HelmholtzHexProblem(m, bcs=True, nq=nq)
g = prob.jac_loop._prepare()["cw"].src.tp
for _ in range(2):
    prob.assemble_jacobian()
ev = [(Event(), Event()) for _ in range(5)]
for e in ev:
    prob.assemble_jacobian(events=e)
_lib.call("fd_device_sync")
ms = float(np.median([a.elapsed_ms(b) for a, b in ev]))
fl = prob.ALGO_FLOPS_PER_CELL * m.ncells
print(f"Q{degree} n={n} nq={nq} panel<={panel} chunk={chunk}: tiles {g['tiles']} = {g['col_splits']} x {g['col_tiles']}, "
      f"{g['matrix_groups']} groups of {g['matrix_threads']} lanes per cell: {ms:.3f} ms = {fl / ms / 1e9 / bench.FP64_MFMA_PEAK_TFLOPS:.3f} of the MFMA peak", flush=True)
