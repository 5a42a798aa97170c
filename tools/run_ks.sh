cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M2D_BRANCH_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_ks -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof > $R/gpurun_out/r2_ks.log 2>&1
find $R/gpurun_out/r2_ks -name "*kernel_trace.csv" -delete
python3 $R/tools/kstats.py $R/gpurun_out/r2_ks 24 60
