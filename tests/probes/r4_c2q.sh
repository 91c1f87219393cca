cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do python3 bench.py --cpu-budget 0 --no-live-pmc 2>/dev/null | python3 tests/probes/bench_fields.py c2; done
