cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 1 0; do
  ENLSIP_GN_SB_FORM_HINTS=$v python3 bench.py --cpu-budget 0 --steps 8 2>/dev/null | python3 tests/probes/bench_fields.py hints=$v
done; done
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
