#!/bin/bash
# split-K reduce with four columns per lane (gemm_splitk_reduce4_kernel) vs one: config 4 (42 reduce launches per step), and the headline as a control
out=gpurun_out/r03_reduce_vec4.txt; : > $out
run() { echo "## ${@:2} (config $1)" >> $out; env "${@:2}" timeout -k 10 200 python bench.py --config $1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], 'enc_fwd', s.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'))" >> $out || exit 1; }
for rep in 1 2; do
run 4 BBBP_GEMM_REDUCE_VEC4=1
run 4 BBBP_GEMM_REDUCE_VEC4=0
done
run 3 BBBP_GEMM_REDUCE_VEC4=1
run 3 BBBP_GEMM_REDUCE_VEC4=0
cat $out
