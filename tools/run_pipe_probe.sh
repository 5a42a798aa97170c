ms() { python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])"; }
B="python3 bench.py --no-cpu-baseline --no-prof --steps 32 --warmup 16"
M2D_CAPTURE_FORK=1 timeout 300 python -m pytest tests/test_gpu_full_size.py -x -q -k "graph" 2>&1 | tail -5
for b in 8 64; do
echo "batch$b eager: $($B --batch $b 2>/dev/null | ms)  graphs: $($B --batch $b --graphs on 2>/dev/null | ms)  graphs+fork: $(M2D_CAPTURE_FORK=1 $B --batch $b --graphs on 2>&1 | ms)"
done
