#!/bin/bash
# tools/evidence_r5_final.sh -- the last GPU action of round 5, on the tree after the shapes added behind tools/evidence_r5.sh (periodic
# columns, interior facets, WRITE / MIN / MAX beside staged arguments, transposed vector scatter): smoke(), the whole GPU suite, the
# default bench line (timed), and the 8-rank one-device rehearsals of bench.py --gpus 8.  Kernel trace and PMC summaries of the benchmark
# kernels are those of tools/evidence_r5.sh (profiles/r5z_*): their code did not change.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
touch /tmp/r5zz_marker
python -c "from firedrake_amd import forms; print(len(forms.precompile_all()), 'code objects')"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warning\|amdgpu" | tail -2 | tee gpurun_out/r5zz_smoke.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|setattr\|_float_to_str" | tail -12 > gpurun_out/r5zz_gputests_tail.txt
tail -4 gpurun_out/r5zz_gputests_tail.txt
S=$(date +%s); python bench.py --steps 20 --warmup 5 > gpurun_out/r5zz_bench_line.json 2> gpurun_out/r5zz_bench_line.err; echo "bench.py wall $(( $(date +%s) - S )) s" >> gpurun_out/r5zz_bench_line.err
tail -2 gpurun_out/r5zz_bench_line.err; head -c 600 gpurun_out/r5zz_bench_line.json; echo
for part in slabs blocks; do
  FDHIP_FORCE_DEVICE=0 FDHIP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 1 --partition $part > gpurun_out/r5zz_rehearsal_8ranks_${part}_one_device.json \
    2> gpurun_out/r5zz_rehearsal_${part}.err
  head -c 300 gpurun_out/r5zz_rehearsal_8ranks_${part}_one_device.json; echo
done
mkdir -p gpurun_out/r5zz_cache
find firedrake_amd/_cache -type f -newer /tmp/r5zz_marker \( -name "*.hsaco" -o -name "*.res.json" \) -exec cp {} gpurun_out/r5zz_cache/ \;
ls gpurun_out/r5zz_cache | wc -l
