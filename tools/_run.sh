export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python3 $R/tools/steady.py 16 > /tmp/steady.log 2>&1
grep -v rocprofv3 /tmp/steady.log | tail -3
python tools/fill_timeline.py /tmp/trace 16 > gpurun_out/fill_timeline.txt 2>&1
python tools/body_timeline.py /tmp/trace > gpurun_out/body_timeline.txt 2>&1
head -45 gpurun_out/fill_timeline.txt
