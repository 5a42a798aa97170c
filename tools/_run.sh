R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06g
cd $R
python3 bench.py --dump-shapes $O/${TAG}_shapes_c3.csv > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err
python3 bench.py --config c2 --no-cpu-baseline --parity-check --dump-shapes $O/${TAG}_shapes_c2.csv > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench_c3.err
python3 bench.py --config c4 --no-cpu-baseline --dump-shapes $O/${TAG}_shapes_c4.csv > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench_c3.err
python3 bench.py --config c5 --no-cpu-baseline --dump-shapes $O/${TAG}_shapes_c5.csv > $O/${TAG}_bench_c5.json 2>> $O/${TAG}_bench_c3.err
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof --no-other-configs"
M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $ARGS > $O/${TAG}_prof.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_steady -- python3 $R/tools/steady.py 16 > $O/${TAG}_steady.log 2>&1
python3 $R/tools/tail_count.py $O/${TAG}_steady 16 > $O/${TAG}_launch_census.txt 2>&1
rm -rf $O/${TAG}_steady
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
cd $R; tail -c 400 $O/${TAG}_bench_c3.json
