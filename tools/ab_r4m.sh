#!/bin/bash
# tools/ab_r4m.sh -- round 4, session m: (1) LDS atomic cost of the P1 wrapper's conflict windows (microbenchmark patterns 200-202);
# (2) tensor-product GPU tests; (3) C3 with and without the single-cell-row stores, action with half tables / five cubes
mkdir -p gpurun_out
./tools/microbench_lds | grep "P1 window\|pat 1 span\|pat 102\|pat 101" > gpurun_out/r4m_microbench_windows.txt; cat gpurun_out/r4m_microbench_windows.txt
timeout 900 python -m pytest tests/test_gpu_q4_hex.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -15 > gpurun_out/r4m_tests.txt; tail -4 gpurun_out/r4m_tests.txt
for rep in 1 2; do
for v in 1 0; do
  echo "== FDHIP_TP_STORE_SINGLE_ROWS=$v rep $rep"
  FDHIP_TP_STORE_SINGLE_ROWS=$v python bench.py --workload c3 --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; a=d.get('roofline_action',{})
print('  jacobian kernel_ms %.4f frac %.4f assemble_ms %.4f frac_assemble %.4f | action ms %.4f frac_valu %.4f' % (r['ms'], r['frac'], r.get('assemble_ms',0), r.get('frac_assemble',0), a.get('ms',0), a.get('frac_valu',0)))"
done; done 2>&1 | tee gpurun_out/r4m_c3_ab.txt
python -c "
import bench, json
print(json.dumps(bench.measure_c3_action(64, 10, 3)))" 2>gpurun_out/r4m_action_n64.err | tee gpurun_out/r4m_action_n64.json
