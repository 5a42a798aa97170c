cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_product_parity.py -m gpu -x -q > gpurun_out/r2_pp_t.log 2>&1; tail -3 gpurun_out/r2_pp_t.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/r2_prof1.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/r2_prof1.log | cut -c1-200
python3 tools/gap_analysis.py gpurun_out/r2_prof1/*/*kernel_trace.csv > gpurun_out/r2_gaps1.log 2>&1; head -5 gpurun_out/r2_gaps1.log
find gpurun_out/r2_prof1 -name "*kernel_trace.csv" -delete
