#!/bin/bash
# config 5 (B = 4096 forward-only): both branches are matrix-pipe bound.  Default since round 3: one stream + out_proj fold under the bf16
# attention kernel; BBBP_SCREEN_OVERLAP=1 BBBP_FOLD_OUTPROJ=1 = the two-stream schedule without that fold
out=gpurun_out/r03_config5_streams_b.txt; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $out || exit 1; }
for rep in 1 2; do
run A=0
run BBBP_SCREEN_OVERLAP=1 BBBP_FOLD_OUTPROJ=1
done
run BBBP_FOLD_OUTPROJ=1
cat $out
