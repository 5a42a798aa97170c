cd $GRAFT_REPO_ROOT
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_gpu_full_size_parity.py -x -q -s --durations=5 > gpurun_out/r2_fs1.log 2>&1; tail -25 gpurun_out/r2_fs1.log
