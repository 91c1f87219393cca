# round 5: XCD-local grid order of the pair far update for few problems with many tiles (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5x}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_robustness.py tests/test_gpu_full_configs.py tests/test_gpu_parity.py -m gpu -q --tb=short -rf -k "lookahead or c4_full or tsqr or c2_full or shapes" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for i in 1 2; do
  for x in 0 1; do
    ENLSIP_GN_XMAP=$x timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_x${x}_$i.err | python3 tests/probes/bench_fields.py xmap $x >> $O/ab.txt
    ENLSIP_GN_XMAP=$x ENLSIP_GN_PAIR=1 timeout -k 10 300 python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2> $O/c4s_x${x}_$i.err | python3 tests/probes/bench_fields.py xmap $x shard-pairs >> $O/ab.txt
  done
done
timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2.err | python3 tests/probes/bench_fields.py c2 >> $O/ab.txt
cat $O/ab.txt
