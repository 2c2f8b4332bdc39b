#!/bin/bash
# In-situ effect of the Winograd conv2 kernels (one GPU): conv2 form x schedule, then the sweep over the CUs their grids take
# while the branches overlap.  Full bench lines -> gpurun_out/wino_insitu.log, summary on stdout.
out=gpurun_out/wino_insitu.log; mkdir -p gpurun_out; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $out || exit 1; }
run BBBP_CONV_WINOGRAD=0
run BBBP_CONV_WINOGRAD=0 BBBP_SINGLE_STREAM=1
run BBBP_CONV_WINOGRAD=3 BBBP_SINGLE_STREAM=1
for cus in 256 224 208 192 176 160 128; do
  run BBBP_CONV_WINOGRAD=3 BBBP_WINO_CUS=$cus
done
run BBBP_CONV_WINOGRAD=1
run BBBP_CONV_WINOGRAD=2
python - <<'PY'
import json
for l in open('gpurun_out/wino_insitu.log'):
    if l.startswith('##'): print(l.strip()); continue
    d = json.loads(l); s = d['roofline']['sections_ms']
    print(f"  {d['value']:.0f} mol/s {d['ms_per_step']:.3f} ms | " + ' '.join(f"{k}={v:.2f}" for k, v in s.items()))
PY
