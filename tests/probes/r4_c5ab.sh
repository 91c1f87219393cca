cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_full_configs.py tests/test_gpu_robustness.py -m gpu -x -q -k "c5 or fused or c3" 2>&1 | tail -2
for i in 1 2 3; do python3 bench.py --config C5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c5; done
python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c3
