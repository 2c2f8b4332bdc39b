#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python3 -m pytest tests -q -m gpu -x > gpurun_out/r04_y_full.log 2>&1; rc=$?
tail -3 gpurun_out/r04_y_full.log
exit $rc
