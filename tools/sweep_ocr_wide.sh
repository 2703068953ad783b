# tools/sweep_ocr_wide.sh -- whole-entity owner-computes-rows as TWO 1024-lane groups per CU (80 KB row blocks) against three 512-lane
# groups (53 KB): FDHIP_OCR_NNZ_ORDERED x ocr_block_threads on the C2 Jacobian
for rep in ${REPS:-1 2}; do for cfg in ${CONFIGS:-"512:4416" "1024:5400" "1024:5800" "1024:6200" "1024:6600"}; do
T=${cfg%%:*}; V=${cfg##*:}
echo -n "lanes=$T FDHIP_OCR_NNZ_ORDERED=$V: "
env FDHIP_LDS_LIMIT=81920 FDHIP_OCR_NNZ_ORDERED=$V FDHIP_DEBUG=1 python -c "
import sys; sys.argv=['bench.py','--steps','20','--warmup','3','--numbering','lexicographic','--variants','','--no-secondary','--cpu-sample','0','--traffic','off']
from firedrake_amd.configuration import configuration as c; c['ocr_block_threads']=$T
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('step %.4f ms  jacobian kernel %.4f ms  frac %.4f' % (d['ms_per_step'], r['ms'], r['frac']), end='  ')"
grep -o "row blocks=[0-9]* instances=[0-9]* (x[0-9.]* entities).*lds=[0-9]*" /tmp/err.txt | tail -1
done; done
