#!/bin/bash
# conv1 split-bf16 forward: round 3's phase-by-phase kernel against the software-pipelined one (alone, B = 512 / 4096, 1 or 2 work-groups per CU),
# then the headline step with the pipelined kernel inside the training plan (BBBP_C1_TRAIN=1) and config 5
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_c1_pipe.txt
: > $O
BBBP_C1_PIPE=1 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv1 or without_mask or ties or persistent" > gpurun_out/r04_c1_pipe_tests.log 2>&1 || { tail -30 gpurun_out/r04_c1_pipe_tests.log; exit 1; }
tail -2 gpurun_out/r04_c1_pipe_tests.log >> $O
for B in 512 4096; do
  for P in 0 1; do for W in 1 2; do
    echo "B=$B BBBP_C1_PIPE=$P BBBP_C1_PER_CU=$W" >> $O
    BBBP_C1_PIPE=$P BBBP_C1_PER_CU=$W python3 tools/bench_conv1.py $B 2>&1 | grep "forward" | sed 's/max|dw.*//' >> $O || exit 1
  done; done
done
for T in 0 1; do
  echo "headline BBBP_C1_PIPE=1 BBBP_C1_TRAIN=$T" >> $O
  BBBP_C1_PIPE=1 BBBP_C1_TRAIN=$T python3 bench.py --no-cpu-baseline > gpurun_out/r04_c1_train$T.log 2>&1 || { tail -5 gpurun_out/r04_c1_train$T.log; exit 1; }
  python3 - gpurun_out/r04_c1_train$T.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], {k:r["sections_ms"][k] for k in ("conv1_fwd","conv2_fwd","encoder_fwd","encoder_bwd","conv1_wgrad")}, {k:r["sections_ms_isolated"][k] for k in ("conv1_fwd",)})
PY
done
for W in 1 2; do
  echo "config 5 BBBP_C1_PIPE=1 BBBP_C1_PER_CU=$W" >> $O
  BBBP_C1_PIPE=1 BBBP_C1_PER_CU=$W python3 bench.py --config 5 --no-cpu-baseline > gpurun_out/r04_c5_pipe$W.log 2>&1 || exit 1
  python3 - gpurun_out/r04_c5_pipe$W.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], {k:r["sections_ms"][k] for k in ("conv1_fwd","conv2_fwd","encoder_fwd")})
PY
done
cat $O
