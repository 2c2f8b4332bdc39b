#!/bin/bash
# Run GPU steps one after another on the gpurun box: a step that FAILS (non-zero, e.g. a red test) does not stop the next one, a step
# that was killed at its time limit (124 / 137) does -- never start GPU work after a hang.  Usage: tools/gpu_steps.sh "cmd1" "cmd2" ...
rc_all=0
for cmd in "$@"; do
  echo "### $cmd"
  bash -o pipefail -c "$cmd"
  rc=$?
  echo "### rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "### killed at its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
