#!/bin/bash
# Collect the per-round evidence on the GPU box: tools/profile_round.sh rNN "3 2 5 4 1"   (writes gpurun_out/<rNN>_*; copy into profiles/)
# Per configuration C: the bench line, the rocprofv3 kernel table of the same command (--no-isolated: every launch in it is an
# in-step one, so its averages are comparable with bench.py's own HIP-event sections) and, for the conv-dominated configurations,
# HBM traffic from separate --pmc passes (FETCH_SIZE, WRITE_SIZE) plus the SQ busy counters of the headline.
set -o pipefail
R=${1:-r02}
CONFIGS=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for C in $CONFIGS; do
  python3 bench.py --config $C > $O/${R}_bench_config$C.log 2>&1 && tail -1 $O/${R}_bench_config$C.log > $O/${R}_bench_config$C.json || exit 1
  SHORT="--steps 10 --warmup 2"; [ $C = 4 ] && SHORT="--steps 4 --warmup 1"; [ $C = 1 ] && SHORT="--steps 4 --warmup 1"; [ $C = 5 ] && SHORT="--steps 6 --warmup 2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats$C -- python3 bench.py --config $C $SHORT --no-cpu-baseline --no-isolated > $O/${R}_stats$C.log 2>&1 || exit 1
  cp $(find $O/${R}_stats$C -name '*kernel_stats.csv' | head -1) $O/${R}_kernel_stats_config$C.csv
  rm -rf $O/${R}_stats$C
  if [ $C = 3 ] || [ $C = 2 ] || [ $C = 5 ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_fetch$C -- python3 bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_fetch$C.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_write$C -- python3 bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_write$C.log 2>&1 || exit 1
    python3 tools/pmc_traffic.py $(find $O/${R}_fetch$C -name '*counter_collection.csv' | head -1) $(find $O/${R}_write$C -name '*counter_collection.csv' | head -1) $O/${R}_pmc_traffic_config$C.json $C
    if [ $C = 3 ]; then
      cp $(find $O/${R}_fetch$C -name '*counter_collection.csv' | head -1) $O/${R}_pmc_fetch_size.csv
      cp $(find $O/${R}_write$C -name '*counter_collection.csv' | head -1) $O/${R}_pmc_write_size.csv
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${R}_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_sq.log 2>&1 || exit 1
      python3 tools/pmc_sq.py $(find $O/${R}_sq -name '*counter_collection.csv' | head -1) $O/${R}_pmc_sq.txt
      rm -rf $O/${R}_sq
    fi
    rm -rf $O/${R}_fetch$C $O/${R}_write$C
  fi
  # the bench line again, now that the traffic file of these kernel sources exists (roofline.traffic is filled from it)
  if [ -f $O/${R}_pmc_traffic_config$C.json ]; then
    mkdir -p profiles && cp $O/${R}_pmc_traffic_config$C.json profiles/
    python3 bench.py --config $C > $O/${R}_bench_config$C.log 2>&1 && tail -1 $O/${R}_bench_config$C.log > $O/${R}_bench_config$C.json || exit 1
  fi
  echo "config $C done"; tail -c 300 $O/${R}_bench_config$C.json; echo
done
ls -la $O | grep ${R}_
