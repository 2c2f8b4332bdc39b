#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f.txt
: > $O
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv" > gpurun_out/r04_f_t1.log 2>&1 || { tail -30 gpurun_out/r04_f_t1.log; echo "CONV TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_f_t1.log >> $O
python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity_sizes.py -q -m gpu > gpurun_out/r04_f_t2.log 2>&1 || { tail -30 gpurun_out/r04_f_t2.log; echo "VARIANT/PARITY TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_f_t2.log >> $O
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-isolated > gpurun_out/r04_f_bench$i.log 2>&1 || { tail -5 gpurun_out/r04_f_bench$i.log; echo FAILED >> $O; }
  python3 - gpurun_out/r04_f_bench$i.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"])
PY
done
python3 tools/time_configs.py 2>/dev/null | grep -i "wide" >> $O
BBBP_CONV_WINOGRAD=224 python3 tools/time_configs.py 2>/dev/null | grep -i "wide" | sed 's/^/[conv2-class stages on the f32 kernels, mask 224] /' >> $O
cat $O
