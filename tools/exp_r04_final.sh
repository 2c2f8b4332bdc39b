#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_final_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r04_final_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 300 python3 bench.py > gpurun_out/r04_final_bench.log 2>&1; rc=$?
grep '^{' gpurun_out/r04_final_bench.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['metric'], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
exit $rc
