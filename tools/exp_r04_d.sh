#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_d.txt
: > $O
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv2 or conv_pool or conv_many" > gpurun_out/r04_d_t1.log 2>&1 || { tail -30 gpurun_out/r04_d_t1.log; echo "CONV TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_d_t1.log >> $O
python3 -m pytest tests/test_gpu_parity_sizes.py tests/test_gpu_round4.py -q -m gpu > gpurun_out/r04_d_t2.log 2>&1 || { tail -30 gpurun_out/r04_d_t2.log; echo "PARITY TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_d_t2.log >> $O
for i in 1 2; do
  python3 bench.py --no-cpu-baseline > gpurun_out/r04_d_bench$i.log 2>&1 || { tail -5 gpurun_out/r04_d_bench$i.log; echo FAILED >> $O; }
  python3 - gpurun_out/r04_d_bench$i.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], {k:r["sections_ms"].get(k) for k in ("conv1_fwd","conv2_fwd","imgfc_fwd","encoder_fwd","imgfc_bwd","conv2_wgrad","conv2_dgrad","conv1_wgrad","encoder_bwd")})
print("   isolated", {k:r["sections_ms_isolated"].get(k) for k in ("conv2_dgrad","encoder_bwd")})
PY
done
python3 tools/step_timeline.py 2>/dev/null | sed -n '/step 2/,/step 3/p' >> $O
python3 bench.py --config 2 --no-cpu-baseline > gpurun_out/r04_d_bench_c2.log 2>&1 && tail -1 gpurun_out/r04_d_bench_c2.log | cut -c1-200 >> $O
python3 bench.py --config 4 --no-cpu-baseline > gpurun_out/r04_d_bench_c4.log 2>&1 && tail -1 gpurun_out/r04_d_bench_c4.log | cut -c1-200 >> $O
cat $O
