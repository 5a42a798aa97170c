for v in 0 1 0 1; do echo "M2D_THIN_LONG=$v"; M2D_THIN_LONG=$v python bench.py --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['whole_cycles']['value'], d['roofline']['frac'])"; done
for v in 0 1 0 1; do echo "C2 M2D_THIN_LONG=$v"; M2D_THIN_LONG=$v python bench.py --config c2 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
