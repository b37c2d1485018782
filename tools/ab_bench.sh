# usage: ab.sh "ENV=VAL" ... : alternates baseline and each setting, 20 steps
run() { env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shard-roofline --no-fp32-like --no-roofline | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); p=j['config']['phase_ms_per_epoch']; print('$*', round(j['value'],1), 'it/s', {k: round(v,2) for k,v in p.items()}, 'sum', round(sum(p.values()),2))
"; }
for rep in 1 2; do for s in "$@"; do run $s; done; done
