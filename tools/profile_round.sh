#!/bin/bash
# Collect the per-round evidence on the GPU box: tools/profile_round.sh rNN   (writes gpurun_out/<rNN>_*; copy into profiles/)
# The kernel table is collected with the branch overlap ON: its durations are in-step (in-situ) ones, like bench.py's sections.
set -o pipefail
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/${R}_bench_n1.log 2>&1 && tail -1 $O/${R}_bench_n1.log > $O/${R}_bench_n1.json || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-isolated > $O/${R}_stats.log 2>&1 || exit 1
cp $(find $O/${R}_stats -name '*kernel_stats.csv' | head -1) $O/${R}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_write.log 2>&1 || exit 1
cp $(find $O/${R}_fetch -name '*counter_collection.csv' | head -1) $O/${R}_pmc_fetch_size.csv
cp $(find $O/${R}_write -name '*counter_collection.csv' | head -1) $O/${R}_pmc_write_size.csv
python3 tools/pmc_traffic.py $O/${R}_pmc_fetch_size.csv $O/${R}_pmc_write_size.csv $O/${R}_pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${R}_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_sq.log 2>&1 || exit 1
python3 tools/pmc_sq.py $(find $O/${R}_sq -name '*counter_collection.csv' | head -1) $O/${R}_pmc_sq.txt
rm -rf $O/${R}_stats $O/${R}_fetch $O/${R}_write $O/${R}_sq
ls -la $O | grep ${R}_
