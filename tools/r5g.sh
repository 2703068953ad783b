#!/bin/bash
# tools/r5g.sh -- round 5, GPU session g: variable-layer extrusion (tests + before/after), the 7-neighbour RCCL-vs-host wire test,
# and the 8-rank one-device rehearsals of bench.py --gpus 8 (slabs, blocks)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_variable_layers.py tests/test_gpu_rccl_self.py tests/test_gpu_mixed_periodic.py tests/test_gpu_parity_random.py tests/test_gpu_pyop2_golden.py -x -q -m gpu 2>&1 \
  | grep -v "Warning\|getlimits\|_float_to_str" | tail -8 > gpurun_out/r5g_tests.txt
cat gpurun_out/r5g_tests.txt
O=gpurun_out/r5g_bench_extruded.txt; : > $O
for v in "" "--variable"; do
  for mode in direct auto; do
    FDHIP_MODE=$mode timeout 600 python tools/bench_extruded.py 128 128 $v --check 2>&1 | grep -v "Warning\|amdgpu.ids" >> $O
  done
done
cat $O
for part in slabs blocks; do
  FDHIP_FORCE_DEVICE=0 FDHIP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 1 --partition $part > gpurun_out/r5g_rehearsal_8ranks_${part}_one_device.json \
    2> gpurun_out/r5g_rehearsal_${part}.err
  tail -3 gpurun_out/r5g_rehearsal_${part}.err; head -c 600 gpurun_out/r5g_rehearsal_8ranks_${part}_one_device.json; echo
done
