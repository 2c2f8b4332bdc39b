#!/bin/bash
# config 4 (and 5, 3) with the 64 x 64 split-bf16 GEMM tile for few-tile deep-K products (default) vs the 128-tile split-K plans
out=gpurun_out/r03_b3s.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py --config $1 --steps 40 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], 'enc_fwd', s.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'), 'qkv_fwd', s.get('qkv_fwd'), 'ffn1_fwd', s.get('ffn1_fwd'), 'outproj_fwd', s.get('outproj_fwd'), 'ffn2_dgrad', s.get('ffn2_dgrad'), 'qkv_dgrad', s.get('qkv_dgrad'))" >> $out || exit 1; }
for rep in 1 2; do
run 4 BBBP_GEMM_B3_SMALL=1
run 4 BBBP_GEMM_B3_SMALL=0
done




cat $out
