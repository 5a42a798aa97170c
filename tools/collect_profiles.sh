# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box via gpurun):
#   kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes.
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-prof"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | cut -c1-300
# keep only the small summaries
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
for d in pmc_fetch_$TAG pmc_write_$TAG; do
python3 - $d <<'PY'
import csv, glob, sys, json, collections
d = sys.argv[1]
f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)[0]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = 'm2d_gemm_kernel' if 'm2d_gemm_kernel' in r['Kernel_Name'] else 'other'
    tot[k] += float(r['Counter_Value']); n[k] += 1
json.dump({"sum_kb": dict(tot), "launches": dict(n)}, open('gpurun_out/%s.json' % d, 'w'))
PY
rm -rf gpurun_out/$d
done
ls gpurun_out/prof_$TAG/*/ gpurun_out/*.json
