#!/bin/bash
# config 4: the optimizer pipelined into the backward pass with a background-sized AdamW (few work-groups, no raised priority)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_r.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d.get('optimizer','')[:60])"; }
echo "config 4 default" >> $O; step timeout -k 10 200 python3 bench.py --config 4 --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
for bg in 0 32 64 128 256; do
  echo "config 4 pipelined step, BBBP_ADAMW_BACKGROUND=$bg" >> $O
  BBBP_BENCH_PIPELINED_STEP=1 BBBP_ADAMW_BACKGROUND=$bg step timeout -k 10 200 python3 bench.py --config 4 --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
done
echo "config 3 pipelined step, BBBP_ADAMW_BACKGROUND=64" >> $O
BBBP_BENCH_PIPELINED_STEP=1 BBBP_ADAMW_BACKGROUND=64 step timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated 2>/dev/null | line >> $O
cat $O
exit 0
