#!/bin/bash
# A/B on one box: forward kernel as committed (A) vs early loads + chunk loop not unrolled (B); the data gradient is the new one in both
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_z.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
P=bbbp-multi-modal-deep-ensemble-framework_amd
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; iso=r.get('sections_ms_isolated',{}); print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k.startswith('conv2')}, {k:round(v,3) for k,v in iso.items() if k.startswith('conv2')})" "$1"; }
for v in A B A B; do
  if [ $v = A ]; then F=""; else F="-DB3_EARLY_LOADS=1 -DB3_CHUNK_NOUNROLL=1"; fi
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC $F -I include -c $P/csrc/conv_b3.hip -o $P/build/conv_b3.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libbbbp_hip.so $P/build/*.o || exit 1
  echo "== variant $v $F" >> $O
  step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line headline >> $O
  step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline 2>/dev/null | line config5 >> $O
  step timeout -k 10 200 python3 bench.py --config 2 --no-cpu-baseline 2>/dev/null | line config2 >> $O
done
step timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k conv 2>&1 | tail -1 >> $O
cat $O
exit 0
