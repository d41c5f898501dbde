for rep in 1 2 3; do for p in 2 3; do
python bench.py --no-sub --cpu-seconds 0 --steps 20 --warmup 5 --pipeline $p 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('K=20 depth $p',d['value'],d['ms_per_step'],d['ms_per_step_min'],d['ms_per_step_max'])"
done; done
for p in 2 3; do
python bench.py --no-sub --cpu-seconds 0 --steps 1000 --warmup 100 --pipeline $p 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('K=1000 depth $p',d['value'],d['ms_per_step'],d['ms_per_step_min'],d['ms_per_step_max'])"
done
