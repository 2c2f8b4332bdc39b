#!/bin/bash
# MLP trainer: shared RandomState streams on the host; Adam-state loads all in flight (variant 0) / also issued before the product (variant 1)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_o.txt; : > $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping" | tee -a $O; exit $rc; fi; }
P=bbbp-multi-modal-deep-ensemble-framework_amd
for v in 0 1; do
  if [ $v = 1 ]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DMLP_PRELOAD=1 -I include -c $P/csrc/mlp.hip -o $P/build/mlp.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libbbbp_hip.so $P/build/*.o || exit 1
  fi
  echo "== MLP_PRELOAD=$v" >> $O
  step timeout -k 10 300 python3 -m pytest tests/test_gpu_mlp.py -q -m gpu > gpurun_out/r04_o_tests$v.log 2>&1; tail -1 gpurun_out/r04_o_tests$v.log >> $O
  step timeout -k 10 300 python3 tools/mlp_phases.py 4 >> $O 2>&1
  step timeout -k 10 300 python3 bench.py --config 1 --no-cpu-baseline >> $O 2>&1
done
grep -v amdgpu.ids $O | cut -c1-260
exit 0
