cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tr in 0 256; do python3 bench.py --config C3 --cpu-budget 0 --tile-rows $tr 2>/dev/null | python3 tests/probes/bench_fields.py c3 tile=$tr; done
for b in 2048 4096; do python3 bench.py --config C3 --cpu-budget 0 --batch $b 2>/dev/null | python3 tests/probes/bench_fields.py c3 batch=$b; done
