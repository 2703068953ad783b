#!/bin/bash
# tools/r5l.sh -- round 5, GPU session l: the whole GPU suite on the tree with 64-bit row starts
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -40 | cut -c1-300 > gpurun_out/r5l_gputests_tail.txt
cat gpurun_out/r5l_gputests_tail.txt
