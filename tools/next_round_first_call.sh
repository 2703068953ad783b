#!/bin/bash
# tools/next_round_first_call.sh -- the measurements round 3 ran out of GPU minutes for, one box, ~6 minutes:
#   1. FDHIP_FLUSH_BATCH 4 | 8 on the C2 Jacobian (both numberings) and on the un-hinted CG2 share
#   2. limiter + traffic counters of the un-hinted row-sliced CG2 Jacobian (every pass under timeout)
#   3. the default bench line of the tree
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
{ timeout 300 tools/ab_env.sh FDHIP_FLUSH_BATCH "4 8"; timeout 120 tools/ab_c5.sh FDHIP_FLUSH_BATCH "4 8"; } > gpurun_out/r4a_ab_flush_batch.txt 2>&1
timeout 500 tools/pmc_c5_r3.sh lexicographic > /dev/null 2>&1; cp gpurun_out/r3_pmc_c5_lexicographic.txt gpurun_out/r4a_pmc_c5_lexicographic.txt 2>/dev/null
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r4a_bench_line.json 2> gpurun_out/r4a_bench_line.err
cat gpurun_out/r4a_ab_flush_batch.txt; head -c 600 gpurun_out/r4a_bench_line.json
