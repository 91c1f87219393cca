cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 1 2; do ENLSIP_GN_FACTOR_NW4=$v python3 bench.py --cpu-budget 0 --steps 6 2>/dev/null | python3 tests/probes/bench_fields.py nw4 $v; done; done
