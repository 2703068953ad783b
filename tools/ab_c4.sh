#!/bin/bash
# tools/ab_c4.sh "<bench args>" VAR=VAL[,VAR=VAL] ... -- like ab.sh for `bench.py --workload c4`
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' ')
  out=$(env $envs python $R/bench.py --workload c4 $ARGS 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
cfg, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    per = " | ".join("%s %.3f ms (%.2f)" % (r["kernel"].replace("wrap_dg_adv_", ""), r["ms"], r["frac"]) for r in d["roofline_per_loop"])
    print("%-45s step %.3f ms | %s | %.3g DoFs/s" % (cfg, d["ms_per_step"], per, d["value"]))
except Exception as e:
    print(cfg, "FAILED", e, out[:200])
PY
done
