#!/bin/bash
# kernel table of the graph-captured wide/deep step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out
export BBBP_WIDE_GRAPH=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_wdg -- python3 tools/bench_wide_deep.py 256 20 > $O/r04_wdg.log 2>&1 || { tail -5 $O/r04_wdg.log; exit 1; }
cp $(find $O/r04_wdg -name '*kernel_stats.csv' | head -1) $O/r04_kernel_stats_wide_deep_graph.csv
python3 tools/trace_timeline.py $(find $O/r04_wdg -name '*kernel_trace.csv' | head -1) > $O/r04_wdg_timeline.txt 2>&1
python3 - <<'P' > $O/r04_wdg_streams.txt 2>&1
import csv, glob, collections
f = glob.glob('gpurun_out/r04_wdg/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 5% of the trace = a few replayed steps: per queue busy time and span
t1 = int(rows[-1]['End_Timestamp']); t0 = t1 - 30_000_000
sel = [r for r in rows if int(r['Start_Timestamp']) >= t0]
byq = collections.defaultdict(lambda: [0, 0])
for r in sel:
    q = r.get('Queue_Id'); byq[q][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); byq[q][1] += 1
print('last 30 ms of the trace:', len(sel), 'kernels')
for q, (busy, n) in byq.items():
    print('queue', q, 'busy %.2f ms' % (busy / 1e6), n, 'kernels')
# union busy time
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in sel)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print('GPU busy (union) %.2f ms of 30' % (u / 1e6))
P
rm -rf $O/r04_wdg
cat $O/r04_wdg_streams.txt; tail -2 $O/r04_wdg.log
