#!/bin/bash
# sparse weight gradient on 32 x 32 maps (the variant's 128 -> 256 stage) + the graph-captured wide/deep step
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_k.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_variants.py tests/test_gpu_round4.py -q -m gpu -k "conv or wide or graph or multi or capturable" > gpurun_out/r04_k_tests.log 2>&1; tail -12 gpurun_out/r04_k_tests.log | tee -a $O
BBBP_WIDE_GRAPH=1 step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
BBBP_WIDE_GRAPH=1 BBBP_C2_WGRAD_SPARSE_WAVES=4 step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
step timeout -k 10 200 python3 tools/bench_wide_deep.py >> $O 2>&1; tail -2 $O
exit 0
