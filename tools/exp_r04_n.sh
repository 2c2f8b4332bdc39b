#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_n.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 300 python3 -m pytest tests/test_gpu_mlp.py -q -m gpu > gpurun_out/r04_n_tests.log 2>&1; tail -3 gpurun_out/r04_n_tests.log | tee -a $O
step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1
step timeout -k 10 300 python3 bench.py --config 1 --no-cpu-baseline >> $O 2>&1
grep -v amdgpu.ids $O | cut -c1-330
exit 0
