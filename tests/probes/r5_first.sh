# round 5, first GPU call: the whole -m gpu suite, then C4 with the deep look-ahead against the round-4 schedule, then its timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5a}; mkdir -p $O
echo "[1] pytest -m gpu"; timeout -k 10 900 python3 -m pytest tests -m gpu -q --maxfail=20 --tb=short -rf > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 15 $O/pytest.log
echo "[2] C4 A/B"
for i in 1 2; do
  for d in 0 1; do
    ENLSIP_GN_LA_DEEP=$d timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_deep${d}_$i.err | python3 tests/probes/bench_fields.py deep $d >> $O/c4_ab.txt
  done
done
cat $O/c4_ab.txt
echo "[3] trace"; timeout -k 10 400 bash tests/probes/trace_c4.sh $O/c4trace > $O/trace.log 2>&1; tail -n 80 $O/trace.log
