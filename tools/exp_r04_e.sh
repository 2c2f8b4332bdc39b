#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_e.txt
: > $O
python3 -m pytest tests/test_gpu_parity_sizes.py -q -m gpu -k "adamw" > gpurun_out/r04_e_t.log 2>&1; tail -1 gpurun_out/r04_e_t.log >> $O
for PR in 0 3 1 0 3; do
  echo "BBBP_CONV_BWD_PRIO=$PR" >> $O
  BBBP_CONV_BWD_PRIO=$PR python3 bench.py --no-cpu-baseline --no-isolated > gpurun_out/r04_e_bench$PR.log 2>&1 || { tail -5 gpurun_out/r04_e_bench$PR.log; echo FAILED >> $O; }
  python3 - gpurun_out/r04_e_bench$PR.log >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], {k:r["sections_ms"].get(k) for k in ("imgfc_bwd","conv2_wgrad","conv2_dgrad","conv1_wgrad","encoder_bwd","encoder_fwd")})
PY
done
for PR in 0 3; do echo "timeline BBBP_CONV_BWD_PRIO=$PR" >> $O; BBBP_CONV_BWD_PRIO=$PR python3 tools/step_timeline.py 2>/dev/null | sed -n '/step 2/,/step 3/p' >> $O; done
cat $O
