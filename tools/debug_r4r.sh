#!/bin/bash
mkdir -p gpurun_out
python tools/debug_pack.py 2>&1 | grep -v Warning | tee gpurun_out/r4r_debug_pack.txt
FDHIP_DEBUG=1 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --traffic off > gpurun_out/r4r_bench.json 2> gpurun_out/r4r_bench.err
grep "variant" gpurun_out/r4r_bench.err | sort | uniq -c | tee gpurun_out/r4r_variants.txt
python -c "
import json; d=json.load(open('gpurun_out/r4r_bench.json')); print(d['jit_compiles']); print(d['setup_s'], d['steps_to_amortise_setup'])"
