#!/bin/bash
# tools/ab.sh "<bench args>" VAR=VAL[,VAR=VAL] ... -- run bench.py under several env settings, print one line each
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' ')
  out=$(env $envs python $R/bench.py $ARGS 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
cfg, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print("%-45s step %.3f ms | res %.3f ms frac %.3f | jac %.3f ms frac %.3f | %.3g DoFs/s" % (cfg, d["ms_per_step"], d["roofline_residual"]["ms"], d["roofline_residual"]["frac"], d["roofline_jacobian"]["ms"], d["roofline_jacobian"]["frac"], d["value"]))
except Exception as e:
    print(cfg, "FAILED", e, out[:200])
PY
done
