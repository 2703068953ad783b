"""tools/time_action.py [n] -- kernel time of the Q4 action wrapper at n (default 32; median of 30), for same-box A/Bs of the template and
as the command of the PMC passes at n = 64 (matrix-free problem: the matrix would not fit the 32-bit CSR index range)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from firedrake_amd import forms, mesh as fmesh, _lib
from firedrake_amd.device import Event
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
prob = forms.HelmholtzQ4Problem(m, bcs=(n <= 32), matrix=(n <= 32))
for _ in range(3):
    prob.assemble_action()
ts = []
for _ in range(30):
    ev = (Event(), Event())
    prob.assemble_action(events=ev)
    _lib.call("fd_device_sync")
    ts.append(ev[0].elapsed_ms(ev[1]))
print("action kernel ms, n = %d (FDHIP_CFLAGS=%s): median %.4f min %.4f" % (n, os.environ.get("FDHIP_CFLAGS", ""), np.median(ts), min(ts)))
