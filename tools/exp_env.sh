#!/bin/bash
# bench under different environment settings: tools/exp_env.sh "A=1 B=2" "A=3" ...   ("-" = no extra env)
for cfg in "$@"; do
  for ss in 0 1; do
    echo "== $cfg single_stream=$ss"
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    env $envs BBBP_SINGLE_STREAM=$ss python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print(d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in r['sections_ms'].items()}, r.get('conv2_dgrad_isolated_clock'))"
  done
done
