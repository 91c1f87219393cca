cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C4
python3 bench.py --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C2
python3 -m pytest tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -3
