#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_m.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1
MLP_ONLY=32 step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1
grep -v amdgpu.ids $O | grep -v "cycles / mini"
exit 0
