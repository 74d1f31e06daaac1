cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $O/pytest3.log 2>&1
tail -6 $O/pytest3.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
