#!/usr/bin/env python
"""tools/power_probe.py [c2|c5] [jacobian|residual] [seconds] -- engine clock and board power while ONE benchmark kernel runs back to back
(rocm-smi sampled from a side thread every 0.25 s).  Answers whether a kernel is held by the power cap: the roofline fractions of
DESIGN.md assume nothing about the clock, but a kernel that the chip runs at 1.8 GHz instead of 2.4 pays for every wasted byte and
instruction twice -- once in issue slots, once in clock."""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from firedrake_amd import _lib, forms, mesh as fmesh  # noqa: E402
from firedrake_amd.device import Event  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
form = sys.argv[2] if len(sys.argv) > 2 else "jacobian"
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
n, degree = (215, 1) if which == "c2" else (107, 2)
m = fmesh.UnitCubeMesh(n, degrees=(degree,), perturb=0.1, numbering="lexicographic")
prob = forms.PoissonProblem(m, degree, bcs=True)
run = prob.assemble_jacobian if form == "jacobian" else prob.assemble_residual
for _ in range(5):
    run()
_lib.call("fd_device_sync")
samples, stop = [], False


def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        except Exception as exc:          # noqa: BLE001
            samples.append(("error", repr(exc)))
            return
        sclk = re.search(r"sclk clock level[^(]*\((\d+)Mhz\)", out)
        pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", out)
        tj = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", out)
        samples.append((float(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, float(tj.group(1)) if tj else None))
        time.sleep(0.25)


th = threading.Thread(target=sample)
th.start()
t0 = time.perf_counter()
times = []
while time.perf_counter() - t0 < secs:
    ev = [(Event(), Event()) for _ in range(50)]
    for e in ev:
        run(events=e)
    _lib.call("fd_device_sync")
    times += [a.elapsed_ms(b) for a, b in ev]
stop = True
th.join()
print(f"{which} {form}: {len(times)} launches back to back, kernel ms median {np.median(times):.4f} (first 50: {np.median(times[:50]):.4f}, last 50: {np.median(times[-50:]):.4f})")
ok = [s for s in samples if s[0] not in (None, "error")]
if ok:
    print("  sclk MHz: " + " ".join("%d" % s[0] for s in ok))
    print("  power W:  " + " ".join("%s" % (("%d" % s[1]) if s[1] is not None else "-") for s in ok))
    print("  Tj C:     " + " ".join("%s" % (("%d" % s[2]) if s[2] is not None else "-") for s in ok))
else:
    print("  rocm-smi gave no readable samples:", samples[:2])
