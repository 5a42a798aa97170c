# Dev tool: A/B an environment switch on ONE box: bash tools/ab_env.sh VAR [rounds] [bench args...]
# alternates `bench.py` with VAR=0 and VAR=1 and prints ms/step of each run.
VAR=$1; ROUNDS=${2:-3}; shift; shift
for r in $(seq $ROUNDS); do
  for v in 0 1; do
    ms=$(env $VAR=$v python3 bench.py --no-cpu-baseline --no-prof "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "$VAR=$v round $r: $ms ms/step"
  done
done
