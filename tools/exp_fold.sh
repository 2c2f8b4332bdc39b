#!/bin/bash
# out_proj folded into the value projection (csrc/fold.hip) vs the reference's operation order: headline and B = 32 / 64 lines
out=gpurun_out/r03_fold_outproj.txt; : > $out
run() { echo "## ${@:2} (args: $1)" >> $out; env "${@:2}" timeout -k 10 150 python bench.py $1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']; i=r.get('sections_ms_isolated',{})
print(d['ms_per_step'], d['value'], r['kernel'], r['ms_per_launch'], r['frac'], 'enc_fwd', s.get('encoder_fwd'), 'alone', i.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'), 'alone', i.get('encoder_bwd'))" >> $out || exit 1; }
for rep in 1 2; do
run "--config 3" BBBP_FOLD_OUTPROJ=1
run "--config 3" BBBP_FOLD_OUTPROJ=0
done
run "--batch 32" BBBP_FOLD_OUTPROJ=1
run "--batch 32" BBBP_FOLD_OUTPROJ=0
run "--batch 64" BBBP_FOLD_OUTPROJ=1
run "--batch 64" BBBP_FOLD_OUTPROJ=0
cat $out
