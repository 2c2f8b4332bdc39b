out=gpurun_out/direct_sweep.log; : > $out
run() { echo "## $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $out || exit 1; }
run A=0
run BBBP_GEMM_DIRECT_T=2
run BBBP_GEMM_DIRECT_KS=1
run BBBP_GEMM_DIRECT_KS=4
run BBBP_GEMM_DIRECT_T=2 BBBP_GEMM_DIRECT_KS=1
run BBBP_WINO_SIDE_CUS=56
run BBBP_WINO_SIDE_CUS=72
run BBBP_CONV_PER_CU=3
python - <<'PY'
import json
for l in open('gpurun_out/direct_sweep.log'):
    if l.startswith('##'): print(l.strip()); continue
    d = json.loads(l); s = d['roofline']['sections_ms']
    print(f"  {d['value']:.0f} mol/s {d['ms_per_step']:.3f} ms | " + ' '.join(f"{k}={v:.2f}" for k, v in s.items()))
PY
