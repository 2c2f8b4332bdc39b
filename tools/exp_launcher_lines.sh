#!/bin/bash
# the launcher / rank-count evidence lines on a one-GPU box (control flow, not scaling): two gloo ranks time-slicing the card, the
# exact-global-batch mode on one rank and on two gloo ranks, and the per-rank shard of a strong-scaled global batch 512 over 8 GPUs
O=gpurun_out
BBBP_BENCH_BACKEND=gloo timeout -k 10 240 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/n2.log 2>$O/n2.err && tail -1 $O/n2.log > $O/r03_bench_n2_gloo_one_gpu.json || { tail -5 $O/n2.err; exit 1; }
timeout -k 10 240 python3 bench.py --exact-batch --no-cpu-baseline > $O/ex1.log 2>&1 && tail -1 $O/ex1.log > $O/r03_bench_exact_batch_n1.json || { tail -5 $O/ex1.log; exit 1; }
BBBP_BENCH_BACKEND=gloo timeout -k 10 240 python3 bench.py --gpus 2 --exact-batch --steps 5 --warmup 2 > $O/ex2.log 2>$O/ex2.err && tail -1 $O/ex2.log > $O/r03_bench_exact_batch_n2_gloo_one_gpu.json || { tail -5 $O/ex2.err; exit 1; }
timeout -k 10 240 python3 bench.py --scaling strong --batch 64 --no-cpu-baseline > $O/ss.log 2>&1 && tail -1 $O/ss.log > $O/r03_bench_strong_shard_b64.json || { tail -5 $O/ss.log; exit 1; }
python3 - <<'PY'
import json
for f in ("r03_bench_n2_gloo_one_gpu", "r03_bench_exact_batch_n1", "r03_bench_exact_batch_n2_gloo_one_gpu", "r03_bench_strong_shard_b64"):
    d = json.load(open(f"gpurun_out/{f}.json")); r = d.get("rccl") or {}
    print(f, d["n_gpus"], d["ms_per_step"], round(d["value"], 1), d["scaling"], r.get("world"), r.get("backend"), r.get("distinct_devices"), r.get("collectives_per_step"), r.get("launched_by"))
PY
