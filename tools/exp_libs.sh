#!/bin/bash
# compare builds of the library: tools/exp_libs.sh lib1.so lib2.so ...   ("default" = the in-tree build)
for lib in "$@"; do
  if [ "$lib" != default ]; then export BBBP_LIB=$PWD/$lib; else unset BBBP_LIB; fi
  for ss in 0 1; do
    echo "== lib=$lib single_stream=$ss"
    BBBP_SINGLE_STREAM=$ss python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['sections_ms'].items()})"
  done
done
