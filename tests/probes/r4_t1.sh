cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_robustness.py -m gpu -x -q -k "form_hints" 2>&1 | tail -15
