# round 5, second GPU call: the tests that failed, C4 deep look-ahead A/B with sC / sD in their own priority class, timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5b}; mkdir -p $O
echo "[1] pytest (subset)"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_problems.py -m gpu -q --maxfail=20 --tb=short -rf -k "extreme or fast_reflector or lookahead" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 8 $O/pytest.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_robustness.py tests/test_gpu_full_configs.py -m gpu -q --tb=short -rf -k "lookahead or c4_full" > $O/pytest2.log 2>&1; echo "rc=$?" >> $O/pytest2.log; tail -n 5 $O/pytest2.log
echo "[2] C4 A/B"
for i in 1 2; do
  for d in 0 1; do
    ENLSIP_GN_LA_DEEP=$d timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_deep${d}_$i.err | python3 tests/probes/bench_fields.py deep $d >> $O/c4_ab.txt
  done
done
cat $O/c4_ab.txt
echo "[3] trace"; timeout -k 10 400 bash tests/probes/trace_c4.sh $O/c4trace > $O/trace.log 2>&1; tail -n 75 $O/trace.log
