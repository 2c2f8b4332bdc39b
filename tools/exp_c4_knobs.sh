#!/bin/bash
# config 4 (F = 2048) under tuning knobs: one bench line per setting -> gpurun_out/r03_c4_knobs.txt
out=gpurun_out/r03_c4_knobs.txt; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], r['kernel'], r['ms_per_launch'], 'enc_fwd', s.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'))" >> $out || exit 1; }
run A=1
run BBBP_ATTN_BWD1=0
run BBBP_GEMM_SPLIT_X10=10
run BBBP_GEMM_SPLIT_X10=30
run BBBP_GEMM_SPLIT_FULL=1
run BBBP_GEMM_SPLIT_X10=10 BBBP_GEMM_SPLIT_FULL=1
cat $out
