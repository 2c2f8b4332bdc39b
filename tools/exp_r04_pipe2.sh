#!/bin/bash
# conv2 forward, software-pipelined one-work-group-per-CU form (BBBP_C2_PIPE=1) against the two-work-group kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_pipe2.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; iso=r.get('sections_ms_isolated',{}); print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k in ('conv2_fwd','encoder_fwd','conv1_fwd')}, {k:round(v,3) for k,v in iso.items() if k in ('conv2_fwd',)})" "$1"; }
BBBP_C2_PIPE=1 step timeout -k 10 400 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_sizes.py -q -m gpu -k "conv or eval or screening or forward" > gpurun_out/r04_pipe2_tests.log 2>&1; tail -4 gpurun_out/r04_pipe2_tests.log | tee -a $O
for v in 0 1 0 1; do
  echo "== BBBP_C2_PIPE=$v" >> $O
  BBBP_C2_PIPE=$v step timeout -k 10 300 python3 tools/bench_conv2.py 512 2>&1 | grep "split-bf16 B=512" | cut -c1-120 >> $O
  BBBP_C2_PIPE=$v step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line headline >> $O
  BBBP_C2_PIPE=$v step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline 2>/dev/null | line config5 >> $O
done
cat $O
exit 0
