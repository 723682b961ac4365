cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pe5; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
