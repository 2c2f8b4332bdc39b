#!/bin/bash
# A/B on one box: conv2 forward / data gradient with (1) / without (0) the next stage's loads placed inside the MFMA block
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_x.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
P=bbbp-multi-modal-deep-ensemble-framework_amd
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; iso=r.get('sections_ms_isolated',{}); print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k.startswith('conv2')}, {k:round(v,3) for k,v in iso.items() if k.startswith('conv2')})" "$1"; }
for v in 1 0 1 0; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DB3_EARLY_LOADS=$v -I include -c $P/csrc/conv_b3.hip -o $P/build/conv_b3.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libbbbp_hip.so $P/build/*.o || exit 1
  echo "== B3_EARLY_LOADS=$v" >> $O
  step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line headline >> $O
  step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline 2>/dev/null | line config5 >> $O
  step timeout -k 10 200 python3 bench.py --config 2 --no-cpu-baseline 2>/dev/null | line config2 >> $O
done
cat $O
exit 0
