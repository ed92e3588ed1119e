for c in 4 8 16 32 64; do
  python bench.py --mode intrafrl --intra-chunks $c --no-cpu-baseline --steps 100 --warmup 20 --prewarm-seconds 0.5 > /tmp/o.json 2>/dev/null
  python -c "
import json; o=json.load(open('/tmp/o.json')); print($c, round(o['ms_per_step'],3), round(o['roofline']['frac'],4), {k: round(v,2) for k,v in o['stages_ms'].items()})"
done
