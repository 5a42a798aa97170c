# Dev tool: A/B an environment switch on ONE box: bash tools/ab_env.sh VAR A B [rounds] [bench args...]
# alternates `bench.py` with VAR=A and VAR=B and prints ms/step of each run and the means.
VAR=$1; A=$2; B=$3; ROUNDS=${4:-3}; shift; shift; shift; shift
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    ms=$(env $VAR=$v python3 bench.py --no-cpu-baseline --no-prof "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "$VAR=$v round $r: $ms ms/step"
  done
done | tee /tmp/ab_env.$$ 
python3 - <<PY
import re,collections
d=collections.defaultdict(list)
for l in open("/tmp/ab_env.$$"):
    m=re.match(r"(\S+) round \d+: ([\d.]+)",l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k,v in d.items(): print(k,"mean %.3f min %.3f n %d"%(sum(v)/len(v),min(v),len(v)))
PY
