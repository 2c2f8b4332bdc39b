#!/bin/bash
# backward half: data gradient in the pipelined one-work-group-per-CU form (co-running) x weight gradient in the 4-wave (co-running) / 8-wave (blocking) form
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_dpipe.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k in ('conv2_dgrad','conv2_wgrad','encoder_bwd','conv1_wgrad','imgfc_bwd')}, {k:round(v,3) for k,v in r.get('sections_ms_isolated',{}).items() if k in ('conv2_dgrad','conv2_wgrad')})" "$1"; }
BBBP_C2_DGRAD_PIPE=1 step timeout -k 10 400 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_sizes.py -q -m gpu -k "conv or B512 or 512 or gradient" > gpurun_out/r04_dpipe_tests.log 2>&1; tail -3 gpurun_out/r04_dpipe_tests.log | tee -a $O
for rep in 1 2; do
step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "A default" >> $O
BBBP_C2_DGRAD_PIPE=1 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "B dgrad pipelined" >> $O
BBBP_C2_DGRAD_PIPE=1 BBBP_C2_WGRAD_SPARSE_WAVES=8 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "C dgrad pipelined + wgrad 8 waves" >> $O
done
BBBP_C2_WGRAD_SPARSE_WAVES=8 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "D wgrad 8 waves only" >> $O
cat $O
exit 0
