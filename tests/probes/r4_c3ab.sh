cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_full_configs.py -m gpu -x -q -k "c3" 2>&1 | tail -2
for i in 1 2; do for v in 1 0; do ENLSIP_GN_JQ1_ROWS2=$v python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c3 rows2=$v; done; done
