# solves/s of the default C2 bench over the batch size, round-4 library (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for b in 256 384 512 640 768 1024; do
  python3 bench.py --cpu-budget 0 --no-roofline --no-live-pmc --batch $b --steps 6 2>/dev/null | python3 tests/probes/bench_fields.py batch $b
done; done
