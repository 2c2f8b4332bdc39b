#!/bin/bash
# screening (config 5): one stream vs two streams with conv1 at one / two work-groups per CU
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_u.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], round(d['value']))"; }
for rep in 1 2; do
echo "one stream" >> $O; step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
echo "two streams, conv1 1 per CU" >> $O; BBBP_SCREEN_OVERLAP=1 step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
echo "two streams, conv1 2 per CU" >> $O; BBBP_SCREEN_OVERLAP=1 BBBP_SCREEN_C1_PER_CU=2 step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
done
cat $O
exit 0
