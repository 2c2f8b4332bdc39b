#!/bin/bash
# A/B two builds of the library on the whole training step: tools/exp_ab.sh build_ab/libX.so [env assignments...]
alt=$1; shift
for lib in "" "$alt" "" "$alt"; do
  if [ -n "$lib" ]; then export BBBP_LIB=$PWD/$lib; else unset BBBP_LIB; fi
  echo "== lib=${lib:-default} $@"
  env "$@" python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['sections_ms'].items()})"
done
