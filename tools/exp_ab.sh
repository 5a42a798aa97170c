# Dev tool: A/B an environment switch on ONE box - per-layer times (tools/conv_layer_time.py) and bench.py ms/step.
#   gpurun -- bash tools/exp_ab.sh VAR A B TAG [rounds] [bench args...]      (values A and B of VAR)
VAR=$1; A=$2; B=$3; TAG=$4; ROUNDS=${5:-3}; shift; shift; shift; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in $A $B; do
  echo "== $VAR=$v" >> $OUT/layers.txt
  env $VAR=$v python3 tools/conv_layer_time.py >> $OUT/layers.txt 2>&1
done
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    ms=$(env $VAR=$v python3 bench.py --no-cpu-baseline --no-prof "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "$VAR=$v round $r: $ms ms/step" >> $OUT/ab.txt
  done
done
grep -v amdgpu.ids $OUT/layers.txt; cat $OUT/ab.txt
