cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 0 16 32 64; do ENLSIP_GN_LA_RESERVE=$r python3 bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C4 reserve=$r; done
