#!/bin/bash
# conv2 forward / data gradient with the next stage's loads placed inside the MFMA block
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_w.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 400 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k conv > gpurun_out/r04_w_tests.log 2>&1; tail -3 gpurun_out/r04_w_tests.log | tee -a $O
step timeout -k 10 300 python3 tools/bench_conv2.py 512 2>&1 | grep -v amdgpu.ids | grep "B=512" | cut -c1-300 | tee -a $O
step timeout -k 10 300 python3 tools/bench_conv2.py 4096 2>&1 | grep -v amdgpu.ids | grep "split-bf16 B=4096" | cut -c1-300 | tee -a $O
for i in 1 2; do step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('headline', d['ms_per_step'], round(d['value']))" | tee -a $O; done
step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config5', d['ms_per_step'], round(d['value']))" | tee -a $O
exit 0
