"""Markdown table of a round's collected bench lines (profiles/<round>_bench_config{1..5}.json) for DESIGN.md section 6.
usage: python tools/round_table.py r04"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
rows = []
for c in (3, 2, 4, 5, 1):
    p = os.path.join(ROOT, "profiles", f"{R}_bench_config{c}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p)); r = d.get("roofline") or {}
    tr = r.get("traffic")
    cpu = (d.get("cpu_baseline") or {}).get("value")
    dom = f"{r.get('kernel')}: {r.get('ms_per_launch')} ms x {r.get('launches_per_step', 1)} in-step, {r.get('achieved')} {r.get('unit')}, frac {r.get('frac')}"
    if r.get("ms_per_launch_isolated"):
        dom += f" (alone {r['ms_per_launch_isolated']} ms, frac {r.get('frac_isolated')})"
    rows.append(f"| {c} | {d['config']['workload'][:70]}... | **{d['ms_per_step']}** | {d['value']:.0f} | {dom} | {('%.1f MB' % (tr / 1e6)) if tr else '—'} | {cpu} |")
print("| config | workload | ms / step | molecules (or sample-visits) / s | dominant kernel | PMC traffic per launch | CPU baseline |")
print("|---|---|---|---|---|---|---|")
print("\n".join(rows))
p = os.path.join(ROOT, "profiles", f"{R}_bench_config3.json")
if os.path.exists(p):
    e = json.load(open(p)).get("roofline_encoder")
    if e:
        print("\nroofline_encoder (config 3):", {k: e[k] for k in ("achieved", "peak", "frac", "gflop_per_step", "kernel_ms_per_step", "launches_per_step", "layernorm_ms_per_step", "chain_ms")})
