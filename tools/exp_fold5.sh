#!/bin/bash
# config 5 (B = 4096 forward-only): out_proj fold on the fused bf16 attention path; in_proj on the tiled instead of the direct GEMM
out=gpurun_out/r03_fold_config5.txt; : > $out
run() { echo "## ${@:2} (args: $1)" >> $out; env "${@:2}" timeout -k 10 200 python bench.py $1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']; i=r.get('sections_ms_isolated',{})
print(d['ms_per_step'], d['value'], r['kernel'], r['ms_per_launch'], r['frac'], 'enc_fwd', s.get('encoder_fwd'), 'alone', i.get('encoder_fwd'))" >> $out || exit 1; }
for rep in 1 2; do
run "--config 5" BBBP_FOLD_OUTPROJ=1
run "--config 5" BBBP_FOLD_OUTPROJ=0
done
run "--config 5" BBBP_FOLD_OUTPROJ=1 BBBP_GEMM_DIRECT=5
cat $out
