#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_b.txt
: > $O
export BBBP_C1_PIPE=1
for K in "0 0" "1 0" "0 1" "1 1"; do
  set -- $K
  echo "training test BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2" >> $O
  BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2 python3 -m pytest tests/test_gpu_training.py -q -m gpu -k "faithful" > gpurun_out/r04_b_t_$1_$2.log 2>&1
  tail -1 gpurun_out/r04_b_t_$1_$2.log >> $O; grep "AssertionError: (" gpurun_out/r04_b_t_$1_$2.log | cut -c1-200 >> $O
done
for K in "0 1" "1 1"; do
  set -- $K
  echo "timeline BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2" >> $O
  BBBP_LN_ABSORB=$1 BBBP_C1_TRAIN=$2 python3 tools/step_timeline.py 2>/dev/null | sed -n '/step 2/,/step 3/p' >> $O
done
python3 tools/exp_b3db_r2.py > gpurun_out/r04_b3db_r2.log 2>&1 || { tail -20 gpurun_out/r04_b3db_r2.log; echo "B3DB FAILED" >> $O; }
cat gpurun_out/r04_b3db_r2.txt >> $O
BBBP_LN_ABSORB=0 BBBP_C1_TRAIN=1 python3 tools/exp_b3db_r2.py > gpurun_out/r04_b3db_r2_ln0.log 2>&1; tail -16 gpurun_out/r04_b3db_r2_ln0.log | head -8 >> $O
cat $O
