#!/bin/bash
# conv1 split-bf16 forward: where the time is (BBBP_C1_EXP: 1 no output stores, 2 no MFMAs, 4 no stage loads), per batch size
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_c1_ablate.txt
: > $O
for B in 512 4096; do
  for E in 0 1 2 4 3 6 7; do
    echo "B=$B BBBP_C1_EXP=$E" >> $O
    BBBP_C1_EXP=$E python3 tools/bench_conv1.py $B 2>&1 | grep "forward" >> $O || exit 1
  done
done
cat $O
