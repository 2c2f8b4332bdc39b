#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 tools/adamw_ulp_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
