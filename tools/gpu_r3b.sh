cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 3000 python -m pytest tests/ -m gpu -x -q > gpurun_out/r3d/tests_full.log 2>&1
tail -3 gpurun_out/r3d/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
