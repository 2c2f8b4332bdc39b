#!/bin/bash
# in-situ A/B of the direct GEMM path (env knob BBBP_GEMM_DIRECT: 0 off, 1 all, 3 only K<=192, 5 only 16x16 wave tiles, 7 both)
for cfg in "0 4" "1 4" "3 4" "1 2" "1 8" "5 4"; do
  set -- $cfg
  for ss in 0 1; do
    echo "== BBBP_GEMM_DIRECT=$1 KS=$2 BBBP_SINGLE_STREAM=$ss"
    BBBP_GEMM_DIRECT=$1 BBBP_GEMM_DIRECT_KS=$2 BBBP_SINGLE_STREAM=$ss python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['sections_ms'].items()})"
  done
done
