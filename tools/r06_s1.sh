cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_timeline; mkdir -p $O
python tools/timeline.py 2>&1 | grep -v amdgpu.ids > $O/timeline.txt; cat $O/timeline.txt | head -90
