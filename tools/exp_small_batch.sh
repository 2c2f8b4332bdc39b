#!/bin/bash
out=gpurun_out/r03_small_batch4.txt; : > $out
run() { env "${@:2}" timeout -k 10 120 python bench.py --batch $1 --steps 300 --warmup 50 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" >> $out || exit 1; }
for rep in 1 2 3; do for B in 32 64 128; do
  run $B BBBP_FUSED_ENCODER=0
  run $B BBBP_FUSED_ENCODER=2
done; done
sort $out
