#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export BBBP_BENCH_BACKEND=gloo
timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_n2_gloo_one_gpu.log 2>&1; echo "rc $?"
grep '^{' gpurun_out/r04_bench_n2_gloo_one_gpu.log | tail -1 > gpurun_out/r04_bench_n2_gloo_one_gpu.json
python3 -c "
import json; d=json.load(open('gpurun_out/r04_bench_n2_gloo_one_gpu.json')); print(d['n_gpus'], d['ms_per_step'], d['value'], json.dumps(d['rccl']['comm']))"
