"""tools/time_action.py -- kernel time of the Q4 action wrapper at n = 32 (median of 30), for same-box A/Bs of the template."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from firedrake_amd import forms, mesh as fmesh, _lib
from firedrake_amd.device import Event
m = fmesh.make_extruded_hex_mesh(32, 32, 4, perturb=0.1)
prob = forms.HelmholtzQ4Problem(m, bcs=True)
for _ in range(3):
    prob.assemble_action()
ts = []
for _ in range(30):
    ev = (Event(), Event())
    prob.assemble_action(events=ev)
    _lib.call("fd_device_sync")
    ts.append(ev[0].elapsed_ms(ev[1]))
print("action kernel ms (FDHIP_CFLAGS=%s): median %.4f min %.4f" % (os.environ.get("FDHIP_CFLAGS", ""), np.median(ts), min(ts)))
