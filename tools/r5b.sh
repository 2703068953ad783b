#!/bin/bash
# tools/r5b.sh -- round 5, GPU session b (one box): the table flushes with their loads hoisted above the barrier (FDHIP_FLUSH_PRELOAD)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_records.py tests/test_gpu_sliced.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -3 > gpurun_out/r5b_tests.txt
{ timeout 300 tools/ab_env.sh FDHIP_FLUSH_PRELOAD "0 1"; } > gpurun_out/r5b_ab_p1_preload.txt 2>&1
timeout 200 tools/ab_c5.sh FDHIP_FLUSH_PRELOAD "0 1" > gpurun_out/r5b_ab_c5_preload.txt 2>&1
{ timeout 120 python tools/phase_times.py c2; timeout 120 python tools/phase_times.py c5; } > gpurun_out/r5b_phase_times.txt 2>&1
cat gpurun_out/r5b_tests.txt gpurun_out/r5b_ab_p1_preload.txt gpurun_out/r5b_ab_c5_preload.txt gpurun_out/r5b_phase_times.txt
