for rep in 1 2 3; do
for v in "--graphs off" "--graphs on"; do
  timeout 600 python bench.py --no-cpu-baseline --config c2 --no-other-configs --no-prof $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['whole_cycles']['ms_per_step'])"
done; done
M2D_STEP_TIMES=1 timeout 600 python bench.py --no-cpu-baseline --config c2 --no-other-configs --no-prof --graphs on 2>&1 >/dev/null | grep "step ms"
