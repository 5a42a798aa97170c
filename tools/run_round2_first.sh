cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t1.log 2>&1; tail -3 gpurun_out/r2_t1.log
M2D_BRANCH_OVERLAP=0 python bench.py --no-cpu-baseline > gpurun_out/r2_b1_off.log 2>&1; tail -1 gpurun_out/r2_b1_off.log | cut -c1-400
M2D_BRANCH_OVERLAP=1 python bench.py --no-cpu-baseline > gpurun_out/r2_b1_on.log 2>&1; tail -1 gpurun_out/r2_b1_on.log | cut -c1-400
M2D_BRANCH_OVERLAP=0 python bench.py --no-cpu-baseline --no-prof > gpurun_out/r2_b1_off2.log 2>&1; tail -1 gpurun_out/r2_b1_off2.log | cut -c1-200
M2D_BRANCH_OVERLAP=1 python bench.py --no-cpu-baseline --no-prof > gpurun_out/r2_b1_on2.log 2>&1; tail -1 gpurun_out/r2_b1_on2.log | cut -c1-200
python bench.py --gpus 2 --backend gloo --same-device --batch 16 --steps 8 --warmup 8 --no-cpu-baseline --no-prof > gpurun_out/r2_b1_dp2.log 2>&1; tail -1 gpurun_out/r2_b1_dp2.log | cut -c1-300
M2D_BRANCH_OVERLAP=1 python tools/debug_bench.py > gpurun_out/r2_dbg1.log 2>&1; grep "host enqueue" gpurun_out/r2_dbg1.log
