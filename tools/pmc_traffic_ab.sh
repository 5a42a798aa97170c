# Engine HBM traffic per step with a switch off / on (separate PMC passes, counters in their own runs):
#   bash tools/pmc_traffic_ab.sh M2D_TILE_MAP r04a   -> gpurun_out/<tag>_pmc_traffic_<VAR>0.json / ..._<VAR>1.json
R=$GRAFT_REPO_ROOT
VAR=$1
TAG=${2:-r04}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof"
for v in 0 1; do
  export $VAR=$v
  M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_$v -- python3 $ARGS > /dev/null 2>&1
  M2D_BRANCH_OVERLAP=0 M2D_GEN_PIPELINE=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_$v -- python3 $ARGS > /dev/null 2>&1
  (cd $R && python3 tools/pmc_summary.py $O/${TAG}_pmc_fetch_$v $O/${TAG}_pmc_write_$v > $O/${TAG}_pmc_traffic_${VAR}$v.json)
  rm -rf $O/${TAG}_pmc_fetch_$v $O/${TAG}_pmc_write_$v
done
unset $VAR
cat $O/${TAG}_pmc_traffic_${VAR}0.json $O/${TAG}_pmc_traffic_${VAR}1.json
