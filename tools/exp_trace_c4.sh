#!/bin/bash
# per-dispatch kernel trace of config 4 (F = 2048): which GEMM launches are slow, by grid size and position in the stream
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/r03_trace4 -- python3 bench.py --config ${CFG:-4} --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $O/r03_trace4.log 2>&1 || exit 1
f=$(find $O/r03_trace4 -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' > $O/r03_trace_config${CFG:-4}.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step only: find the last adamw launch and take the dispatches between the previous one and it
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:44]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  grid {r['Grid_Size_X']:>7}x{r['Grid_Size_Y']:>5}x{r['Grid_Size_Z']:>4}  wg {r['Workgroup_Size_X']:>4}  {name}")
PY
rm -rf $O/r03_trace4
tail -3 $O/r03_trace_config${CFG:-4}.txt
