#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_pipe4.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
P=bbbp-multi-modal-deep-ensemble-framework_amd
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], round(d['value']), {k:round(v,3) for k,v in r['sections_ms'].items() if k in ('conv2_fwd','encoder_fwd','conv1_fwd')}, {k:round(v,3) for k,v in r.get('sections_ms_isolated',{}).items() if k in ('conv2_fwd',)})" "$1"; }
for rep in 1 2; do
step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "headline default" >> $O
BBBP_C2_PIPE=1 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "headline PIPE=1" >> $O
BBBP_C2_PIPE=0 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "headline PIPE=0" >> $O
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DB3P_ABLATE_DX -I include -c $P/csrc/conv_b3.hip -o $P/build/conv_b3.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libbbbp_hip.so $P/build/*.o || exit 1
echo "== ablation: dx taps without their LDS reads (wrong results)" >> $O
BBBP_C2_PIPE=1 step timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | line "ablated PIPE=1 headline" >> $O
BBBP_C2_PIPE=1 step timeout -k 10 200 python3 bench.py --config 5 --no-cpu-baseline 2>/dev/null | line "ablated PIPE=1 config5" >> $O
cat $O
exit 0
