#!/bin/bash
# chunk sweep of the learn || Adam+Polyak pipeline (diagnostic)
for c in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chunks $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chunks', $c, round(d['value']), 'env-steps/s', round(d['ms_per_step'],2), 'ms/step', {k: round(v,2) for k,v in d['stages_ms'].items()})"
done
