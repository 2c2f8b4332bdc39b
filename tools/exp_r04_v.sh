#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
BBBP_B3_PROBE=1 timeout -k 10 300 python3 tools/bench_conv2.py 512 2>&1 | grep -v amdgpu.ids | cut -c1-400
BBBP_B3_PROBE=1 timeout -k 10 300 python3 tools/bench_conv2.py 4096 2>&1 | grep -v amdgpu.ids | grep -i "split-bf16\|b3" | cut -c1-400
