#!/bin/bash
out=gpurun_out/r03_mask.txt; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-isolated $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['sections_ms']
print(d['ms_per_step'], d['value'], r['kernel'], r['ms_per_launch'], {k: s.get(k) for k in ('conv1_fwd','conv2_fwd','encoder_fwd','conv2_wgrad','conv2_dgrad','conv1_wgrad','encoder_bwd')})" >> $out || exit 1; }
EXTRA="--config 3"; run BBBP_CONV_WINOGRAD=60; run BBBP_CONV_WINOGRAD=124 BBBP_C1_PER_CU=1; run BBBP_CONV_WINOGRAD=60; run BBBP_CONV_WINOGRAD=124 BBBP_C1_PER_CU=1
cat $out
