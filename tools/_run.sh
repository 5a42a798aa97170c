for v in 0 1 0 1; do echo "M2D_UPSAMPLE_FLAT=$v"; M2D_UPSAMPLE_FLAT=$v python bench.py --config c5 --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm']; print(d['value'], d['ms_per_step'], {k:h[k].get('frac') for k in ('upsample2_fwd','upsample2_bwd') if k in h})"; done
