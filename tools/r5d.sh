#!/bin/bash
# tools/r5d.sh -- round 5, GPU session d (one box): the row-sliced wrapper with persistent workgroups and software-pipelined blocks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_sliced.py tests/test_gpu_records.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -4 > gpurun_out/r5d_tests.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or cg2 or share" 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -4 >> gpurun_out/r5d_tests.txt
timeout 300 tools/ab_c5.sh FDHIP_OCRS_PERSISTENT "0 1" > gpurun_out/r5d_ab_c5_persistent.txt 2>&1
timeout 300 tools/ab_c5.sh FDHIP_OCRS_PERSISTENT "0 1" --numbering tiled > gpurun_out/r5d_ab_c5_persistent_tiled.txt 2>&1
cat gpurun_out/r5d_tests.txt gpurun_out/r5d_ab_c5_persistent.txt gpurun_out/r5d_ab_c5_persistent_tiled.txt
