#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for FA in 7 1; do
BBBP_FLASH_ATTENTION=$FA BBBP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_c5_stats -- python3 bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-isolated > $O/r03_c5_stats_fa$FA.log 2>&1 || exit 1
cp $(find $O/r03_c5_stats -name '*kernel_stats.csv' | head -1) $O/r03_kernel_stats_c5_single_stream_fa$FA.csv
rm -rf $O/r03_c5_stats
done
