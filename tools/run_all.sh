cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2_all_t.log 2>&1; tail -4 gpurun_out/r2_all_t.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
