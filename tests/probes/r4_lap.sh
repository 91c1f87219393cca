cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_robustness.py -m gpu -x -q -k "lookahead" 2>&1 | tail -3
for i in 1 2; do for la in 0 x; do
  if [ $la = 0 ]; then export ENLSIP_GN_LOOKAHEAD=0; else unset ENLSIP_GN_LOOKAHEAD; fi
  python3 bench.py --batch 1 --steps 20 --cpu-budget 0 --no-roofline 2>/dev/null | python3 tests/probes/bench_fields.py single la=$la
  python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py shard la=$la
  python3 bench.py --batch 4 --steps 20 --cpu-budget 0 --no-roofline 2>/dev/null | python3 tests/probes/bench_fields.py batch4 la=$la
done; done
