# tools/sweep_ocrs_wide.sh -- row-sliced owner-computes-rows (CG2 share): four 256-lane groups per CU on 40 KB blocks against two
# 512-lane groups on 80 KB blocks, with the sliced LDS budget decoupled from the staged loops' (configuration ocrs_lds_limit)
for rep in ${REPS:-1 2}; do for cfg in ${CONFIGS:-"256:4096:0" "512:8192:81920" "512:7680:81920" "512:8704:90112"}; do
T=$(echo $cfg | cut -d: -f1); V=$(echo $cfg | cut -d: -f2); L=$(echo $cfg | cut -d: -f3)
echo -n "lanes=$T FDHIP_OCRS_NNZ=$V ocrs_lds_limit=$L: "
env FDHIP_OCRS_NNZ=$V FDHIP_OCRS_BLOCK_THREADS=$T FDHIP_DEBUG=1 python -c "
import sys; sys.argv=['bench.py','--workload','c5','--n','107','--numbering','lexicographic','--steps','20','--warmup','3','--cpu-sample','0','--traffic','off']
from firedrake_amd.configuration import configuration as c; c['ocrs_lds_limit']=$L
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('step %.4f ms  jacobian %.4f ms (%.4f)  residual %.4f ms' % (d['ms_per_step'], r['ms'], r['frac'], q['ms']), end='  ')"
grep -o "lds=[0-9]*" /tmp/err.txt | sort | uniq -c | tr '\n' ' '; echo
done; done
