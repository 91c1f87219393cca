# pivoted QR of more than 512 rows: launch-per-step head + register blocks (default) against launch-per-step to the end
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_robustness.py -m gpu -x -q > gpurun_out/r4h/pytest.log 2>&1 || { tail -30 gpurun_out/r4h/pytest.log; exit 1; }
tail -3 gpurun_out/r4h/pytest.log
for i in 1 2; do for v in 1 0; do
  ENLSIP_GN_QRCP_HYBRID=$v python3 bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py hybrid=$v
  ENLSIP_GN_QRCP_HYBRID=$v python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py hybrid=$v shard
done; done
