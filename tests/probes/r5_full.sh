# round 5: the whole -m gpu suite, then one measurement round (tests/probes/measure_round.sh) — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5m}; mkdir -p $O
echo "[0] pytest -m gpu"; timeout -k 10 900 python3 -m pytest tests -m gpu -q --maxfail=20 --tb=short -rf > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 6 $O/pytest.log
timeout -k 10 900 bash tests/probes/measure_round.sh $O > $O/measure.log 2>&1; tail -n 30 $O/measure.log
for f in c2 c3 c5 c4 c4_shard32768; do python3 tests/probes/bench_fields.py $f < $O/$f.json; done
