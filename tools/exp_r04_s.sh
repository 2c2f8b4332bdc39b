#!/bin/bash
# N-rank rehearsal on one GPU (gloo, two ranks sharing the card): the launcher, the comm diagnostics, every config's N > 1 path; plus smoke()
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_s.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
export BBBP_BENCH_BACKEND=gloo
echo "== smoke" >> $O; step timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O 2>&1
echo "== config 3 --gpus 2" >> $O; step timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_n2_gloo_one_gpu.log 2>&1; echo "rc $?" >> $O; tail -1 gpurun_out/r04_bench_n2_gloo_one_gpu.log > gpurun_out/r04_bench_n2_gloo_one_gpu.json
echo "== config 3 --gpus 2 --exact-batch" >> $O; step timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --exact-batch > gpurun_out/r04_bench_exact_batch_n2_gloo_one_gpu.log 2>&1; echo "rc $?" >> $O; tail -1 gpurun_out/r04_bench_exact_batch_n2_gloo_one_gpu.log > gpurun_out/r04_bench_exact_batch_n2_gloo_one_gpu.json
for c in 1 2 4 5; do
  echo "== config $c --gpus 2" >> $O; step timeout -k 10 300 python3 bench.py --config $c --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_config${c}_n2.log 2>&1; echo "rc $?" >> $O; tail -1 gpurun_out/r04_bench_config${c}_n2.log | cut -c1-400 >> $O
done
python3 - <<'P' >> $O
import json
for f in ("r04_bench_n2_gloo_one_gpu", "r04_bench_exact_batch_n2_gloo_one_gpu"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["n_gpus"], d["ms_per_step"], d["value"], json.dumps(d.get("rccl", {}).get("comm"))[:600])
    except Exception as e:
        print(f, "unreadable:", e)
P
grep -v amdgpu.ids $O | cut -c1-700
exit 0
