#!/bin/bash
# small per-GPU batches (the reference trains at batch 32; a strong-scaled shard of 512 over 8 GPUs is 64): step time by stream / graph mode
out=gpurun_out/r03_small_batch.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 120 python bench.py --batch $1 --steps 200 --warmup 30 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['per_gpu_batch'], d['ms_per_step'], d['value'])" >> $out || exit 1; }
for B in 32 64 128 256; do
  run $B A=1
  run $B BBBP_SINGLE_STREAM=1
  run $B BBBP_GRAPHS=1
  run $B BBBP_SINGLE_STREAM=1 BBBP_GRAPHS=1
done
cat $out
