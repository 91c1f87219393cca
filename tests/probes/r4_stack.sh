cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 2048 4096 8192 16384; do python3 bench.py --rows $m --cols 1024 --cons 0 --batch 1 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py stack $m; done
