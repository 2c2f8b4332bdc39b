#!/bin/bash
# the image FC's backward GEMMs alone (dgrad: NN 512 x 65536 x 128 writes 134 MB; wgrad: TN 128 x 65536 x 512 reads 134 MB), kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_imgfc_gemm.txt; : > $out
run() {  # name, then env assignments, then -- one_gemm args
  name=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    rm -rf gpurun_out/prof_g
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -- python3 tools/one_gemm.py "$@" > gpurun_out/prof_g.log 2>&1 || exit 1
    echo "## $name: $*" >> $out
    python3 - >> $out <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_g/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Name']: print('  ', r['Name'][:70], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3, 1), 'min', round(float(r['MinNs'])/1e3, 1))
PY
  ) || exit 1
}
run xcd_rows_on -- nn 512 65536 128 1
run xcd_rows_off BBBP_GEMM_XCD_ROWS=0 -- nn 512 65536 128 1
run xcd_rows_on_b256 -- nn 256 65536 128 1
run xcd_rows_off_b256 BBBP_GEMM_XCD_ROWS=0 -- nn 256 65536 128 1
run wgrad_default -- tn 128 65536 512 1
rm -rf gpurun_out/prof_g
cat $out
