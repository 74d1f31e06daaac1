cd $GRAFT_REPO_ROOT
python3 tools/diag_stage_probe.py C5 2>&1 | tail -8
