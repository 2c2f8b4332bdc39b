#!/bin/bash
# config 4: work-groups of the fused small-head attention forward (256 = one per head, 512 / 1024 = query blocks of a head split)
out=gpurun_out/r03_attn_parts.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py --config $1 --steps 40 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], 'enc_fwd', s.get('encoder_fwd'), 'attn_fwd', s.get('attn_fwd'), 'attn_bwd', s.get('attn_bwd'))" >> $out || exit 1; }
for rep in 1 2; do
run 4 BBBP_ATTN_FWD_WGS=256
run 4 BBBP_ATTN_FWD_WGS=512
run 4 BBBP_ATTN_FWD_WGS=1024
done
cat $out
