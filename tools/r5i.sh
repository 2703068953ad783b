#!/bin/bash
# tools/r5i.sh -- round 5, GPU session i: the wider tensor descriptors (tests + bench entry), variable layers again, the corner wire test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_tensor_forms.py tests/test_gpu_variable_layers.py tests/test_gpu_q4_hex.py -q -m gpu 2>&1 \
  | grep -v "Warning\|getlimits\|_float_to_str" | tail -25 | cut -c1-600 > gpurun_out/r5i_tests.txt
cat gpurun_out/r5i_tests.txt
timeout 600 python - <<'PY' 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r5i_tensor_forms.txt
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
print(json.dumps(bench.measure_tensor_forms(5, 2), indent=1))
r = bench.measure_c3(32, 5, 2)
print(json.dumps({k: r[k] for k in ("roofline", "roofline_action")}, indent=1))
print(json.dumps(bench.measure_c3_action(64, 10, 2)))
PY
