#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
: > $O/r03_attn_b3_exp.txt
for E in 0 1 2 3 4 8 12 15; do
BBBP_ATTN_B3_EXP=$E BBBP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_c5_stats -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $O/r03_c5_stats_e.log 2>&1 || exit 1
echo "EXP=$E $(grep attn_b3_fwd $(find $O/r03_c5_stats -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4)" >> $O/r03_attn_b3_exp.txt
rm -rf $O/r03_c5_stats
done
cat $O/r03_attn_b3_exp.txt
