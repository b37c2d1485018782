#!/bin/bash
# usage: bash tools/bench_phases.sh [ENV=VAL ...] -- prints value + per-phase ms of a short bench run (no roofline / baseline legs)
env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-shard-roofline --no-fp32-like --no-roofline | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$*', round(j['value'],1), 'it/s', {k: round(v,2) for k,v in j['config']['phase_ms_per_epoch'].items()})
"
