#!/bin/bash
# pipelined conv2 forward selected by the engine for training plans: tests + benches
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_pipe3.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], round(d['value']))" "$1"; }
step timeout -k 10 500 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_ops.py tests/test_gpu_parity_sizes.py tests/test_gpu_variants.py -q -m gpu > gpurun_out/r04_pipe3_tests.log 2>&1; tail -3 gpurun_out/r04_pipe3_tests.log | tee -a $O
for i in 1 2; do
step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line headline >> $O
BBBP_C2_TRAIN=0 step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line headline_C2_TRAIN0 >> $O
done
step timeout -k 10 200 python3 bench.py --config 4 --no-cpu-baseline --no-isolated 2>/dev/null | line config4 >> $O
BBBP_C2_TRAIN=0 step timeout -k 10 200 python3 bench.py --config 4 --no-cpu-baseline --no-isolated 2>/dev/null | line config4_C2_TRAIN0 >> $O
step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | line config5 >> $O
step timeout -k 10 200 python3 bench.py --config 2 --no-cpu-baseline --no-isolated 2>/dev/null | line config2 >> $O
BBBP_WIDE_GRAPH=1 step timeout -k 10 200 python3 tools/bench_wide_deep.py 2>&1 | grep "wide/deep" >> $O
BBBP_WIDE_GRAPH=1 BBBP_C2_PIPE=0 step timeout -k 10 200 python3 tools/bench_wide_deep.py 2>&1 | grep "wide/deep" | sed 's/$/ (BBBP_C2_PIPE=0)/' >> $O
cat $O
exit 0
