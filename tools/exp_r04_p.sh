#!/bin/bash
# full GPU suite (from the test that failed last time on), then (only if green) the round's evidence collection
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python3 -m pytest tests -q -m gpu -x > gpurun_out/r04_p_full.log 2>&1; rc=$?
tail -3 gpurun_out/r04_p_full.log
[ $rc -eq 0 ] || exit $rc
tools/exp_r04_r.sh || exit 1
exec tools/profile_r04.sh
