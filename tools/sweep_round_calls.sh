for R in 8 16 24 32 48 64 128; do RADE_ROUND_CALLS=$R python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print($R, round(d['ms_per_step'],2), r['launches_per_step'], r['per_class_ms'])"; done
