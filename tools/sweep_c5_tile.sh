for v in 768 3072 1536; do
echo "== c5 n=107 lexicographic FDHIP_LOCALITY_TILE=$v"
FDHIP_LOCALITY_TILE=$v timeout 40 python bench.py --workload c5 --n 107 --steps 15 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('  step_ms %.4f  jacobian_ms %.4f  residual_ms %.4f  first_call_s %.3f' % (d['ms_per_step'], r['ms'], q['ms'], d['setup_s']['plans_jacobian_first_call']))"
done
