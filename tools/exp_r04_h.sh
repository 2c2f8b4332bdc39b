#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_h.txt
: > $O
python3 -m pytest tests -q -m gpu -x > gpurun_out/r04_h_full.log 2>&1 || { tail -40 gpurun_out/r04_h_full.log; echo "FULL SUITE FAILED" >> $O; }
tail -1 gpurun_out/r04_h_full.log >> $O
python3 tools/bench_wide_deep.py 2>/dev/null | tail -1 >> $O
BBBP_WIDE_OVERLAP=0 BBBP_CONV_WINOGRAD=224 python3 tools/bench_wide_deep.py 2>/dev/null | tail -1 >> $O
python3 bench.py --no-cpu-baseline --no-isolated > gpurun_out/r04_h_bench.log 2>&1; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_h_bench.log').read().strip().splitlines()[-1]);print('headline',d['ms_per_step'],d['value'])" >> $O
python3 bench.py --config 2 --no-cpu-baseline --no-isolated > gpurun_out/r04_h_bench2.log 2>&1; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_h_bench2.log').read().strip().splitlines()[-1]);print('config2',d['ms_per_step'],d['value'])" >> $O
cat $O
