#!/bin/bash
# conv2's weight gradient on the 2:4 structured-sparse MFMA (4- or 8-wave work-groups) vs the dense split-bf16 kernel
out=gpurun_out/r03_sparse_wgrad.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py --config $1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']; i=r.get('sections_ms_isolated',{})
print(d['ms_per_step'], d['value'], r['kernel'], r['ms_per_launch'], r['frac'], 'conv2_wgrad', s.get('conv2_wgrad'), 'alone', i.get('conv2_wgrad'), 'enc_bwd', s.get('encoder_bwd'))" >> $out || exit 1; }
for rep in 1 2 3; do
run 3 BBBP_C2_WGRAD_SPARSE=1 BBBP_C2_WGRAD_SPARSE_WAVES=4
run 3 BBBP_C2_WGRAD_SPARSE=1 BBBP_C2_WGRAD_SPARSE_WAVES=8
run 3 BBBP_C2_WGRAD_SPARSE=0
done
run 2 BBBP_C2_WGRAD_SPARSE=1 BBBP_C2_WGRAD_SPARSE_WAVES=4
run 2 BBBP_C2_WGRAD_SPARSE=1 BBBP_C2_WGRAD_SPARSE_WAVES=8
run 2 BBBP_C2_WGRAD_SPARSE=0
cat $out
