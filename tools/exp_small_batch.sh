#!/bin/bash
out=gpurun_out/r03_small_batch2.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 120 python bench.py --batch $1 --steps 200 --warmup 30 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['per_gpu_batch'], d['ms_per_step'], d['value'])" >> $out || exit 1; }
for B in 32 64; do
  run $B BBBP_SINGLE_STREAM=1 BBBP_GRAPHS=1 BBBP_GEMM_DIRECT_KS=2
  run $B BBBP_SINGLE_STREAM=1 BBBP_GRAPHS=1 BBBP_GEMM_DIRECT_KS=4
  run $B BBBP_SINGLE_STREAM=1 BBBP_GRAPHS=1 BBBP_GEMM_DIRECT_KS=8
  run $B BBBP_GEMM_DIRECT_KS=8
done
cat $out
