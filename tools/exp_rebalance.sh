#!/bin/bash
# knobs re-checked after the sparse conv2 weight gradient and the out_proj fold moved the balance of the two branches (headline)
out=gpurun_out/r03_rebalance.txt; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $out || exit 1; }
run A=0
run BBBP_C2_WGRAD_SPARSE_WAVES=8
run BBBP_GEMM_DIRECT_KS=4
run BBBP_GEMM_DIRECT_KS=1
run BBBP_B3_PER_CU=1
run A=0
cat $out
