#!/bin/bash
out=gpurun_out/r03_c1_exp.txt; : > $out
for cfg in "0 2" "1 2" "2 2" "7 2"; do set -- $cfg; echo "## EXP=$1 PER_CU=$2" >> $out; BBBP_C1_EXP=$1 BBBP_C1_PER_CU=$2 timeout -k 10 100 python tools/bench_conv1.py 512 2>/dev/null | grep "forward" | cut -c1-100 >> $out || exit 1; done
BBBP_C1_EXP=0 timeout -k 10 100 python tools/bench_conv1.py 4096 2>/dev/null | cut -c1-100 >> $out
cat $out
