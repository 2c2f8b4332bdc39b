#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_g.txt
: > $O
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv" > gpurun_out/r04_g_t1.log 2>&1 || { tail -30 gpurun_out/r04_g_t1.log; echo "CONV TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_g_t1.log >> $O
python3 -m pytest tests/test_gpu_variants.py -q -m gpu > gpurun_out/r04_g_t2.log 2>&1 || { tail -30 gpurun_out/r04_g_t2.log; echo "VARIANT TESTS FAILED" >> $O; }
tail -1 gpurun_out/r04_g_t2.log >> $O
python3 tools/bench_wide_deep.py 2>/dev/null | tail -1 >> $O
BBBP_WIDE_OVERLAP=0 python3 tools/bench_wide_deep.py 2>/dev/null | tail -1 >> $O
BBBP_WIDE_OVERLAP=0 BBBP_CONV_WINOGRAD=224 python3 tools/bench_wide_deep.py 2>/dev/null | tail -1 >> $O
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_wd -- python3 tools/bench_wide_deep.py 256 10 > gpurun_out/r04_wd.log 2>&1
cp $(find gpurun_out/r04_wd -name '*kernel_stats.csv' | head -1) gpurun_out/r04_kernel_stats_wide_deep.csv; rm -rf gpurun_out/r04_wd
head -16 gpurun_out/r04_kernel_stats_wide_deep.csv | cut -c1-170 >> $O
cat $O
