#!/bin/bash
# in-situ effect of the Winograd conv2 kernels under different schedules (one GPU); full bench lines -> gpurun_out/wino_insitu.log
out=gpurun_out/wino_insitu.log; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $out || exit 1; }
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=192
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=208
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=176
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=192 BBBP_RESERVED_CUS=64
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=224 BBBP_RESERVED_CUS=32
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=192 BBBP_RESERVED_CUS=32
run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=192 BBBP_CONV_PER_CU=1
python - <<'PY'
import json
for l in open('gpurun_out/wino_insitu.log'):
    if l.startswith('##'): print(l.strip()); continue
    d = json.loads(l); s = d['roofline']['sections_ms']
    print(f"  {d['value']:.0f} mol/s {d['ms_per_step']:.3f} ms | " + ' '.join(f"{k}={v:.2f}" for k, v in s.items()))
PY
