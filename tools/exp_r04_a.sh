#!/bin/bash
# round 4, batch A: parity of the LayerNorm-absorbing GEMM + the pipelined conv1 forward in training plans, then the headline under the new knobs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_a.txt
: > $O
export BBBP_C1_PIPE=1 BBBP_C1_TRAIN=1
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "layernorm or gemm or conv1" > gpurun_out/r04_a_t1.log 2>&1 || { tail -40 gpurun_out/r04_a_t1.log; echo "OPS TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_a_t1.log >> $O
python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fold.py tests/test_gpu_parity_sizes.py tests/test_gpu_training.py tests/test_gpu_config2.py -q -m gpu > gpurun_out/r04_a_t2.log 2>&1 || { tail -60 gpurun_out/r04_a_t2.log; echo "MODEL TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_a_t2.log >> $O
for K in "0 0" "1 0" "0 1" "1 1" "1 2"; do
  set -- $K
  echo "headline BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2 (BBBP_C1_PIPE=1)" >> $O
  BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2 python3 bench.py --no-cpu-baseline > gpurun_out/r04_a_bench_$1_$2.log 2>&1 || { tail -5 gpurun_out/r04_a_bench_$1_$2.log; exit 1; }
  python3 - gpurun_out/r04_a_bench_$1_$2.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], {k:r["sections_ms"].get(k) for k in ("conv1_fwd","conv2_fwd","imgfc_fwd","encoder_fwd","encoder_bwd","conv1_wgrad","ln_fwd","qkv_fwd","ffn1_fwd","ffn2_fwd")})
e=d.get("roofline_encoder")
if e: print("   encoder:", e["achieved"], "TFLOP/s frac", e["frac"], "launches", e["launches_per_step"], "kernel ms", e["kernel_ms_per_step"], e["chain_ms"])
PY
done
cat $O
# --- batch B additions: the float64-MFMA MLP trainer and the B3DB-scale acceptance run ---
python3 -m pytest tests/test_gpu_mlp.py -x -q -m gpu > gpurun_out/r04_a_t3.log 2>&1 || { tail -40 gpurun_out/r04_a_t3.log; echo "MLP TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_a_t3.log >> $O
python3 bench.py --config 1 > gpurun_out/r04_a_bench_c1.log 2>&1 && tail -1 gpurun_out/r04_a_bench_c1.log | cut -c1-400 >> $O
BBBP_MLP_SCALAR=1 python3 bench.py --config 1 --no-cpu-baseline > gpurun_out/r04_a_bench_c1_scalar.log 2>&1 && tail -1 gpurun_out/r04_a_bench_c1_scalar.log | cut -c1-300 >> $O
python3 tools/exp_b3db_r2.py > gpurun_out/r04_b3db_r2.log 2>&1 || { tail -20 gpurun_out/r04_b3db_r2.log; echo "B3DB FAILED" >> $O; }
cat gpurun_out/r04_b3db_r2.txt >> $O
cat $O
