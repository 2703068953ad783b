for rep in ${REPS:-1 2}; do for v in ${VALUES:-3072 3456 3840 4224 4608 5120}; do
echo -n "FDHIP_OCR_NNZ_ORDERED=$v: "
env FDHIP_OCR_NNZ_ORDERED=$v python bench.py --steps 20 --warmup 3 --numbering lexicographic --variants "" --no-secondary --cpu-sample 0 --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('step %.4f ms  jacobian kernel %.4f ms  frac %.4f  first_call_s %.3f' % (d['ms_per_step'], r['ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
done; done
