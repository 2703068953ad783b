#!/bin/bash
# tools/r5f.sh -- round 5, GPU session f: the whole GPU suite and the default bench line on the tree of the moment
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -8 > gpurun_out/r5f_gputests_tail.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f_bench_line.json 2> gpurun_out/r5f_bench_line.err
cat gpurun_out/r5f_gputests_tail.txt; tail -5 gpurun_out/r5f_bench_line.err; head -c 1500 gpurun_out/r5f_bench_line.json
