cd $GRAFT_REPO_ROOT
PROBE_STEPS=40 python tools/probe_steps2.py 2>&1 | tail -2
