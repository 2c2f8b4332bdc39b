#!/bin/bash
# multi-tensor / capturable AdamW + graph-captured step of the wide/deep variant
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_i.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 400 python3 -m pytest tests/test_gpu_round4.py -q -m gpu > gpurun_out/r04_i_tests.log 2>&1; tail -15 gpurun_out/r04_i_tests.log | tee -a $O
step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
BBBP_WIDE_GRAPH=1 step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
BBBP_WIDE_GRAPH=1 BBBP_WIDE_OVERLAP=0 step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
exit 0
