# Dev tool: like ab_env.sh with any number of values: bash tools/ab_env_n.sh VAR ROUNDS v1 v2 v3 ... [-- bench args]
VAR=$1; ROUNDS=$2; shift; shift
VALS=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS="$VALS $1"; shift; done
[ "$1" = "--" ] && shift
for r in $(seq $ROUNDS); do
  for v in $VALS; do
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
