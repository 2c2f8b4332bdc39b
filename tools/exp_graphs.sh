#!/bin/bash
# HIP-graph replay of the two engine calls (BBBP_GRAPHS=1) against the eager three-stream enqueue, configs 4 / 3 / 2 and small batches
out=gpurun_out/r03_graphs.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py $1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], 'enc_fwd', s.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'))" >> $out || exit 1; }
for g in 0 1; do
run "--config 4 --steps 40 --warmup 8" BBBP_GRAPHS=$g
run "--config 3" BBBP_GRAPHS=$g
run "--config 2" BBBP_GRAPHS=$g
run "--config 5" BBBP_GRAPHS=$g
run "--batch 32 --steps 300 --warmup 50" BBBP_GRAPHS=$g
run "--batch 128 --steps 300 --warmup 50" BBBP_GRAPHS=$g
done
cat $out
