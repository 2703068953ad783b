#!/bin/bash
# tools/evidence_r3.sh -- end-of-round evidence on the final tree (one MI355X): the default bench line, the kernel trace of the
# same command, PMC limiter counters of both C2 kernels under the headline (un-hinted) numbering, and the other configs' lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/r3z_bench_line.json 2> gpurun_out/r3z_bench_line.err
tools/trace.sh r3z --steps 5 --warmup 2 --cpu-sample 0 --variants "" --traffic off > gpurun_out/r3z_trace_summary.txt 2>&1
cp gpurun_out/r3z_trace/t_kernel_stats.csv gpurun_out/r3z_step_kernel_stats.csv 2>/dev/null
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --variants= --traffic off --no-secondary"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r3zpmc_$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/r3zpmc_$name.log 2>&1
done
cd $R
for k in wrap_poisson_p1_tet_jacobian wrap_poisson_p1_tet_residual; do echo "== $k (numbering: lexicographic, no hints)"; python tools/pmc_summary.py $k gpurun_out/r3zpmc_*/; done > gpurun_out/r3z_pmc_summary.txt
rm -rf gpurun_out/r3zpmc_* gpurun_out/r3z_trace
for w in c1 c3 c4; do python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/r3z_bench_$w.json 2>/dev/null; done
head -c 900 gpurun_out/r3z_bench_line.json; echo; cat gpurun_out/r3z_trace_summary.txt | head -12; cat gpurun_out/r3z_pmc_summary.txt | head -40
