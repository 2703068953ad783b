# tools/sweep_ocrs_nnz.sh -- row-sliced owner-computes-rows (CG2 share, n = 107): accumulator entries per row block (FDHIP_OCRS_NNZ)
for rep in ${REPS:-1 2}; do for v in ${VALUES:-3072 3584 3840 4096 4608 5120}; do
echo -n "FDHIP_OCRS_NNZ=$v: "
env FDHIP_OCRS_NNZ=$v FDHIP_DEBUG=1 python bench.py --workload c5 --n 107 --numbering lexicographic --steps 20 --warmup 3 --cpu-sample 0 --traffic off 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('step %.4f ms  jacobian kernel %.4f ms  frac %.4f' % (d['ms_per_step'], r['ms'], r['frac']), end='  ')"
grep -o "lds=[0-9]*" /tmp/err.txt | sort | uniq -c | tr '\n' ' '; echo
done; done
