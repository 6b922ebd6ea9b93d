for cfg in "8 5,8,14" "8 5,7,12" "8 5,10,18" "12 5,8,14" "16 5,8,14" "10 5,8,14" "8 5,5,5"; do set -- $cfg; RADE_ROUND_CALLS=$1 RADE_UNIT_COSTS=$2 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', '$2', round(d['ms_per_step'],2), r['launches_per_step'], r['per_class_ms'])"; done
