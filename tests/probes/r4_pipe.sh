cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in x 1; do
  if [ $v = x ]; then unset ENLSIP_GN_PIPELINE; else export ENLSIP_GN_PIPELINE=1; fi
  python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c3 pipe=$v
  python3 bench.py --config C5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py c5 pipe=$v
done; done
