#!/bin/bash
# tools/check_r4u.sh -- sparsity builder with 2^27-key chunks and a reserved accumulator: parity tests, C2 pattern at full size, setup profile
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity_random.py tests/test_gpu_mixed_periodic.py tests/test_gpu_pyop2_golden.py tests/test_gpu_forms.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -4
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2_against_oracle and lexicographic or c5_per_gpu" 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -3
python tools/setup_profile.py > gpurun_out/r4u_setup_profile.txt 2>&1; grep -n "^==\|fd_csr_from_maps_ex\|fd_memcpy_h2d  " gpurun_out/r4u_setup_profile.txt | head -12
