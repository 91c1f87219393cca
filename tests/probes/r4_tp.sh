cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2_full_size" 2>&1 | tail -3
for i in 1 2; do for v in 1 0; do
  ENLSIP_GN_TRI_PAIR=$v python3 bench.py --cpu-budget 0 --steps 8 2>/dev/null | python3 tests/probes/bench_fields.py tp=$v
done; done
