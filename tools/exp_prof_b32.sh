#!/bin/bash
# kernel durations of one small-batch step with both sliced persistent kernels on (fused-encoder mode 6), one stream
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export BBBP_FUSED_ENCODER=${MODE:-6}
for B in 32 128; do
BBBP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_b${B}_stats -- python3 bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-isolated > $O/r03_b${B}_stats.log 2>&1 || exit 1
cp $(find $O/r03_b${B}_stats -name '*kernel_stats.csv' | head -1) $O/r03_kernel_stats_b${B}_mode${BBBP_FUSED_ENCODER}.csv
rm -rf $O/r03_b${B}_stats
done
