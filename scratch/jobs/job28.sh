cd $GRAFT_REPO_ROOT
PROFILE_FN=_HeteroGatLayer.backward python tools/host_phases.py 2>&1 | grep -v Warn | tail -45
PROFILE_FN=_LinearPair.backward python tools/host_phases.py 2>&1 | grep -v Warn | tail -40
