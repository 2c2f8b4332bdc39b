#!/bin/bash
# the optimizer step pipelined into the backward pass (default) vs one AdamW launch after it
out=gpurun_out/r03_pipelined_step.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py $1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('optimizer','')[:40])" >> $out || exit 1; }
for rep in 1 2 3; do
run "--config 3" BBBP_BENCH_PIPELINED_STEP=1
run "--config 3" BBBP_BENCH_PIPELINED_STEP=0
done
for rep in 1 2; do
run "--config 2" BBBP_BENCH_PIPELINED_STEP=1
run "--config 2" BBBP_BENCH_PIPELINED_STEP=0
run "--config 4 --steps 40 --warmup 8" BBBP_BENCH_PIPELINED_STEP=1
run "--config 4 --steps 40 --warmup 8" BBBP_BENCH_PIPELINED_STEP=0
done
cat $out
