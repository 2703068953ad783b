#!/bin/bash
# tools/ab_trees.sh OLD_DIR [reps] -- same-box A/B of the C2 step between this tree and another checkout of the repository (a git
# worktree with its own libfdhip.so, sharing this tree's code-object cache): alternating runs of the driver's command reduced to
# the headline (no variants, no secondary configs, no CPU leg, no PMC children)
OLD=$1; REPS=${2:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
export FDHIP_CACHE_DIR=$R/firedrake_amd/_cache
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_residual']; j=d['roofline_jacobian']
print('  step %.4f ms  %.3e DoFs/s | residual %.4f ms (%.3f)  assemble %.4f | jacobian %.4f ms (%.3f)  assemble %.4f' % (d['ms_per_step'], d['value'], r['ms'], r['frac'], r['assemble_ms'], j['ms'], j['frac'], j['assemble_ms']))"; }
for i in $(seq $REPS); do
  for tree in "$OLD" "$R"; do
    echo "== $(basename $tree)"
    (cd $tree && python bench.py --steps 20 --warmup 5 --variants "" --no-secondary --traffic off --cpu-sample 0 2>/dev/null | line)
  done
done
