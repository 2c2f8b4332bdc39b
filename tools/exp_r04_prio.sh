#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_prio.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k in ('conv2_fwd','encoder_fwd','conv1_fwd','imgfc_fwd')})" "$1"; }
for rep in 1 2; do for pr in 0 1 2 3; do
BBBP_C2_FWD_PRIO=$pr step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line "BBBP_C2_FWD_PRIO=$pr" >> $O
done; done
cat $O
exit 0
