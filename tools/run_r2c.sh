cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5)
python tools/pipeline_probe.py 1 5 2>&1 | grep "ms/pair\|collapse\|pyr_down\|warp"
python tools/pipeline_probe.py 2 5 2>&1 | grep "ms/pair\|collapse_gather_final"
python tools/pipeline_probe.py 0 5 2>&1 | grep "ms/pair\|collapse_gather_final"
