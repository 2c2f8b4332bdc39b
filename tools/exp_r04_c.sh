#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_c.txt
: > $O
python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_mlp.py tests/test_gpu_training.py tests/test_gpu_overlap_allreduce.py -q -m gpu > gpurun_out/r04_c_t1.log 2>&1 || { tail -60 gpurun_out/r04_c_t1.log; echo "TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_c_t1.log >> $O
for D in 0 1 0 1; do
  echo "headline BBBP_BENCH_DEFER_ADAMW=$D" >> $O
  BBBP_BENCH_DEFER_ADAMW=$D python3 bench.py --no-cpu-baseline --no-isolated > gpurun_out/r04_c_bench_d$D.log 2>&1 || { tail -5 gpurun_out/r04_c_bench_d$D.log; echo FAILED >> $O; }
  python3 - gpurun_out/r04_c_bench_d$D.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], d.get("optimizer_ms_per_step"), {k:r["sections_ms"].get(k) for k in ("conv1_fwd","conv2_fwd","imgfc_fwd","encoder_fwd","encoder_bwd","conv1_wgrad")})
PY
done
python3 bench.py --config 1 > gpurun_out/r04_c_bench_c1.log 2>&1 && tail -1 gpurun_out/r04_c_bench_c1.log | cut -c1-330 >> $O
python3 bench.py --config 5 --no-cpu-baseline > gpurun_out/r04_c_bench_c5.log 2>&1 && tail -1 gpurun_out/r04_c_bench_c5.log | cut -c1-330 >> $O
python3 bench.py --config 2 --no-cpu-baseline > gpurun_out/r04_c_bench_c2.log 2>&1 && tail -1 gpurun_out/r04_c_bench_c2.log | cut -c1-330 >> $O
cat $O
