#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_c1t.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k in ('conv2_fwd','encoder_fwd','conv1_fwd','imgfc_fwd')})" "$1"; }
for rep in 1 2; do
BBBP_C1_TRAIN=1 step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line "C1_TRAIN=1 (default)" >> $O
BBBP_C1_TRAIN=2 step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line "C1_TRAIN=2" >> $O
BBBP_C1_TRAIN=0 step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line "C1_TRAIN=0 (f32 conv1)" >> $O
done
cat $O
exit 0
