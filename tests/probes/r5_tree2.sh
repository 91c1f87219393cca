# round 5: both far tree nodes of a pair in one launch (ENLSIP_GN_TREE2) — parity subset, then same-box A/B on C2 — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5t2}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -q --tb=short -rf -k "c2 or golden or shape_sweep or pairs or randomised or pipelined or batched or mixed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for i in 1 2 3; do
  for x in 0 1; do
    ENLSIP_GN_TREE2=$x timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2_x${x}_$i.err | python3 tests/probes/bench_fields.py tree2 $x >> $O/ab.txt
  done
done
cat $O/ab.txt
