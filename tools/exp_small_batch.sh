#!/bin/bash
# step time of small batches with the launch-per-op schedule (0), the sliced persistent forward (2), backward (4) and both (6)
out=gpurun_out/r03_small_batch5.txt; : > $out
run() { env "${@:2}" timeout -k 10 120 python bench.py --batch $1 --steps 300 --warmup 50 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" >> $out || exit 1; }
for rep in 1 2; do for B in 32 64 128; do
  for m in 0 2 4 6; do run $B BBBP_FUSED_ENCODER=$m; done
done; done
sort $out
