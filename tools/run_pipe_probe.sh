B="python3 bench.py --no-cpu-baseline --no-prof --steps 96 --warmup 16"
export M2D_STEP_TIMES=1
summ() { python3 -c "
import sys
for l in sys.stdin:
    if l.startswith('step ms'):
        v=[float(x) for x in l.split()[2:]]
        s=sorted(v); n=len(v)
        print('n',n,'mean %.3f'%(sum(v)/n),'median %.2f'%s[n//2],' cycle:',' '.join('%.1f'%x for x in v[8:17]))
"; }
timeout 300 python -m pytest tests/test_gpu_full_size.py -x -q -k "pipelined or graph" 2>&1 | tail -3
for r in 1 2 3; do
  echo -n "A   : "; M2D_GEN_PIPELINE=1 $B 2>&1 >/dev/null | summ
  echo -n "off : "; M2D_GEN_PIPELINE=0 $B 2>&1 >/dev/null | summ
done
for c in c4 c5; do
  echo -n "$c A   : "; M2D_GEN_PIPELINE=1 $B --config $c 2>&1 >/dev/null | summ
  echo -n "$c off : "; M2D_GEN_PIPELINE=0 $B --config $c 2>&1 >/dev/null | summ
done
