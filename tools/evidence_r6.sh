#!/bin/bash
# tools/evidence_r6.sh -- end-of-round evidence on the final tree (one MI355X): smoke(), the default bench line (all five configs, N = 1
# anchor of configs[4], CPU baselines, accumulation A/B), kernel traces of C2, of the CG2 share and of configs[4] IN SEPARATE rocprofv3
# runs (so that profiles/ reproduces the P2 averages: the round-5 verdict's evidence item b), of C4 and C3; PMC limiter + traffic
# counters of every benchmark kernel incl. the n = 215 CG2 run (its own counters: item a); the full GPU suite; the 8-rank one-device
# rehearsals of bench.py --gpus 8; and the code objects this box compiled (back into the tree's cache).
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r6z}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
touch /tmp/${T}_marker
python -c "from firedrake_amd import forms; print(len(forms.precompile_all()), 'code objects')"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warning\|amdgpu" | tail -2 | tee gpurun_out/${T}_smoke.txt
S=$(date +%s); python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench_line.err; echo "bench.py wall $(( $(date +%s) - S )) s" >> gpurun_out/${T}_bench_line.err
tail -1 gpurun_out/${T}_bench_line.err; head -c 700 gpurun_out/${T}_bench_line.json; echo
trace() {  # tag, bench args...
  TAG=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_trace_$TAG -o t -- python $R/bench.py "$@" --steps 20 --warmup 5 --cpu-sample 0 --traffic off > $R/gpurun_out/${T}_trace_$TAG.log 2>&1
  cp $R/gpurun_out/${T}_trace_$TAG/t_kernel_stats.csv $R/gpurun_out/${T}_${TAG}_kernel_stats.csv 2>/dev/null
  cd $R
  { echo "== bench.py $* --steps 20 --warmup 5 (rocprofv3 --kernel-trace --stats)"; python tools/trace_summary.py gpurun_out/${T}_${TAG}_kernel_stats.csv 12; } >> gpurun_out/${T}_trace_summary.txt 2>&1
  rm -rf gpurun_out/${T}_trace_$TAG gpurun_out/${T}_trace_$TAG.log
}
rm -f gpurun_out/${T}_trace_summary.txt
trace c2 --variants "" --no-secondary
trace c5share --workload c5 --n 107 --numbering lexicographic
trace c5full --workload c5
trace c4 --workload c4
trace c3 --workload c3
pmc() {  # tag, kernel list, bench args...
  TAG=$1; KERNS=$2; shift 2
  cd /tmp
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_ANY" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${T}pmc_${TAG}_$name -o p -- python $R/bench.py "$@" --steps 3 --warmup 1 --cpu-sample 0 --traffic off > $R/gpurun_out/${T}pmc_${TAG}_$name.log 2>&1
  done
  cd $R
  for k in $(echo $KERNS | tr ',' ' '); do echo "== $k: bench.py $*"; python tools/pmc_summary.py $k gpurun_out/${T}pmc_${TAG}_*/; done
  rm -rf gpurun_out/${T}pmc_${TAG}_*
}
{ pmc c2 wrap_poisson_p1_tet_jacobian,wrap_poisson_p1_tet_residual --variants "" --no-secondary
  pmc c5share wrap_poisson_p2_tet_jacobian,wrap_poisson_p2_tet_residual --workload c5 --n 107 --numbering lexicographic
  pmc c5full wrap_poisson_p2_tet_jacobian,wrap_poisson_p2_tet_residual --workload c5
  pmc c4 wrap_dg_adv_cell,wrap_dg_adv_ext,wrap_dg_adv_int --workload c4; } > gpurun_out/${T}_pmc_summary.txt 2>&1
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${T}pmc_c3_$name -o p -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/gpurun_out/${T}pmc_c3_$name.log 2>&1
done
cd $R
{ for k in wrap_helmholtz_q4_hex_jacobian wrap_helmholtz_q4_hex_action; do echo "== $k: bench.py --workload c3 (n = 32)"; python tools/pmc_summary.py $k gpurun_out/${T}pmc_c3_*/; done; } >> gpurun_out/${T}_pmc_summary.txt 2>&1
rm -rf gpurun_out/${T}pmc_c3_*
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|setattr\|_float_to_str" | tail -12 > gpurun_out/${T}_gputests_tail.txt
tail -3 gpurun_out/${T}_gputests_tail.txt
for part in slabs blocks; do
  FDHIP_FORCE_DEVICE=0 FDHIP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 2 --partition $part > gpurun_out/${T}_rehearsal_8ranks_${part}_one_device.json \
    2> gpurun_out/${T}_rehearsal_${part}.err
  head -c 300 gpurun_out/${T}_rehearsal_8ranks_${part}_one_device.json; echo
done
# the code objects this box compiled (tests + bench): back into the tree's cache so that the next fresh box starts without hipcc
mkdir -p gpurun_out/${T}_cache
find firedrake_amd/_cache -type f -newer /tmp/${T}_marker \( -name "*.hsaco" -o -name "*.res.json" \) -exec cp {} gpurun_out/${T}_cache/ \;
ls gpurun_out/${T}_cache | wc -l
head -30 gpurun_out/${T}_trace_summary.txt
