# round 5: k_caqr_factor_chain — whole suite, then A/B against tree nodes as launches of their own (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5e}; mkdir -p $O
echo "[0] pytest -m gpu"; timeout -k 10 900 python3 -m pytest tests -m gpu -q --maxfail=20 --tb=short -rf > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 8 $O/pytest.log
for i in 1 2; do
  for c in 0 1; do
    ENLSIP_GN_FACTOR_CHAIN=$c timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2_chain${c}_$i.err | python3 tests/probes/bench_fields.py chain $c >> $O/ab.txt
    ENLSIP_GN_FACTOR_CHAIN=$c timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_chain${c}_$i.err | python3 tests/probes/bench_fields.py chain $c >> $O/ab.txt
    ENLSIP_GN_FACTOR_CHAIN=$c timeout -k 10 300 python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2> $O/c4s_chain${c}_$i.err | python3 tests/probes/bench_fields.py chain $c shard >> $O/ab.txt
  done
done
cat $O/ab.txt
