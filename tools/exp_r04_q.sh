#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_gpu_mlp.py -q -m gpu 2>&1 | tail -5
