#!/bin/bash
# Round-4 evidence on the GPU box (writes gpurun_out/r04_*; the summaries are copied to profiles/ afterwards): per configuration the bench line
# + the rocprofv3 kernel table of the same command; PMC traffic for the conv-dominated configurations; a kernel trace of the headline for the
# stall-outlier attribution; the device timeline.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_mlp.py tests/test_gpu_round4.py -q -m gpu > gpurun_out/r04_p_tests.log 2>&1; tail -1 gpurun_out/r04_p_tests.log
tools/profile_round.sh r04 "${1:-3 5 2 1 4}" || exit 1
O=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/r04_trace3 -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-isolated > $O/r04_trace3.log 2>&1 || exit 1
python3 tools/stall_outliers.py $(find $O/r04_trace3 -name '*kernel_trace.csv' | head -1) > $O/r04_stall_outliers.txt 2>&1
python3 tools/trace_timeline.py $(find $O/r04_trace3 -name '*kernel_trace.csv' | head -1) > $O/r04_trace_timeline.txt 2>&1
rm -rf $O/r04_trace3
python3 tools/step_timeline.py > $O/r04_step_timeline.txt 2>&1
python3 tools/exp_b3db_r2.py > $O/r04_b3db_r2.log 2>&1 || tail -5 $O/r04_b3db_r2.log
python3 tools/bench_wide_deep.py > $O/r04_wide_deep.log 2>&1
BBBP_WIDE_GRAPH=1 python3 tools/bench_wide_deep.py >> $O/r04_wide_deep.log 2>&1
BBBP_WIDE_GRAPH=1 BBBP_WIDE_OVERLAP=0 python3 tools/bench_wide_deep.py >> $O/r04_wide_deep.log 2>&1
python3 tools/mlp_phases.py 4 > $O/r04_mlp_phases.log 2>&1
for B in 32 64 128; do python3 bench.py --batch $B --no-cpu-baseline --no-isolated > $O/r04_bench_batch$B.log 2>&1 && tail -1 $O/r04_bench_batch$B.log > $O/r04_bench_batch$B.json; done
tail -5 $O/r04_stall_outliers.txt; tail -3 $O/r04_wide_deep.log
