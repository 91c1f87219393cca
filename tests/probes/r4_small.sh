cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q -k "small or c3 or c5 or hs65 or batch" 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --config C5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c5
python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c3
done
