#!/bin/bash
# graph steps in train_fold + per-op beside-branch wgrad form + MLP phase profile
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_l.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 500 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_variants.py tests/test_gpu_mlp.py -q -m gpu > gpurun_out/r04_l_tests.log 2>&1; tail -12 gpurun_out/r04_l_tests.log | tee -a $O
BBBP_WIDE_GRAPH=1 step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -1 $O
step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1; tail -16 $O
MLP_ONLY=32 step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1; tail -16 $O
exit 0
