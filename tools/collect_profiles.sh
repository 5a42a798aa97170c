# Collect the evidence bench.py's roofline refers to (run on the GPU box via gpurun):
#   bash tools/collect_profiles.sh r02a
# -> gpurun_out/<tag>_*: bench lines (C3 / C4 / C5 per-GPU shapes), per-shape engine + HBM tables,
#    rocprofv3 kernel-trace stats of the same bench command, FETCH_SIZE / WRITE_SIZE PMC passes (separate
#    runs), the MFMA-busy / instruction-mix PMC passes over representative shapes, and the FETCH_SIZE
#    calibration against a known byte count. Copy what should be judged into profiles/.
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out
cd $R
python3 bench.py --dump-shapes $O/${TAG}_shapes_c3.csv > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err
python3 bench.py --config c2 --no-cpu-baseline --parity-check --dump-shapes $O/${TAG}_shapes_c2.csv > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench_c3.err
python3 bench.py --config c4 --no-cpu-baseline --dump-shapes $O/${TAG}_shapes_c4.csv > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench_c3.err
python3 bench.py --config c5 --no-cpu-baseline --dump-shapes $O/${TAG}_shapes_c5.csv > $O/${TAG}_bench_c5.json 2>> $O/${TAG}_bench_c3.err
python3 bench.py --gpus 2 --backend gloo --same-device --batch 16 --steps 8 --warmup 8 --no-cpu-baseline --no-prof > $O/${TAG}_bench_dp2_gloo_1gpu.json 2>> $O/${TAG}_bench_c3.err
cd /tmp && export TMPDIR=/tmp
# (--no-other-configs: the default workload alone - the c2 / c4 / c5 child processes would be traced into the same files)
ARGS="$R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof --no-other-configs"
# one stream (no branch overlap, no generator pipelining): a kernel's duration is then its own, as in bench.py's roofline pass
M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $ARGS > $O/${TAG}_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_overlap -- python3 $ARGS > $O/${TAG}_prof_overlap.log 2>&1
python3 $R/tools/gap_analysis.py $O/${TAG}_prof_overlap > $O/${TAG}_gaps.txt 2>&1
M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $ARGS > /dev/null 2>&1
M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $ARGS > /dev/null 2>&1
# steady-state launch census (no bench prologue / roofline pass / CPU baseline in the trace)
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_steady -- python3 $R/tools/steady.py 16 > $O/${TAG}_steady.log 2>&1
python3 $R/tools/tail_count.py $O/${TAG}_steady 16 > $O/${TAG}_launch_census.txt 2>&1
rm -rf $O/${TAG}_steady
cd $R
find $O/${TAG}_prof $O/${TAG}_prof_overlap -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
# MFMA-busy / instruction mix over representative engine shapes + FETCH_SIZE calibration
bash tools/pmc_shapes.sh $TAG > $O/${TAG}_pmc_shapes.log 2>&1
ls $O | grep "^${TAG}_"
