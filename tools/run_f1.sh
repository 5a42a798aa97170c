cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_lazy_slicing.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r2_f1_t.log 2>&1; tail -5 gpurun_out/r2_f1_t.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy slices', d['ms_per_step'])"; done
