#!/bin/bash
# configs 4 / 3 / 5 with the in-kernel split-K reduction (BBBP_GEMM_FOLD_REDUCE=1, opt-in) and without it -> gpurun_out/r03_c4_fold.txt
out=gpurun_out/r03_c4_fold.txt; : > $out
run() { echo "## $*" >> $out; env "${@:2}" timeout -k 10 150 python bench.py --config $1 --steps 40 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], 'opt', d.get('optimizer_ms_per_step'), r['kernel'], r['ms_per_launch'], 'enc_fwd', s.get('encoder_fwd'), 'enc_bwd', s.get('encoder_bwd'))" >> $out || exit 1; }
for rep in 1 2; do
run 4 BBBP_GEMM_FOLD_REDUCE=1
run 4 BBBP_GEMM_FOLD_REDUCE=0
done
run 3 BBBP_GEMM_FOLD_REDUCE=1
run 3 BBBP_GEMM_FOLD_REDUCE=0
run 5 BBBP_GEMM_FOLD_REDUCE=1
run 5 BBBP_GEMM_FOLD_REDUCE=0
cat $out
