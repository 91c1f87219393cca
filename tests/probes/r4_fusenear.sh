# the pair's narrow update fused into the second panel's factor kernel: parity, then A/B (ENLSIP_GN_FUSE_NEAR=0) on C2 / C3 / C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_robustness.py -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1 || { tail -40 gpurun_out/r4f/pytest.log; exit 1; }
tail -3 gpurun_out/r4f/pytest.log
for i in 1 2; do for v in 1 0; do
  ENLSIP_GN_FUSE_NEAR=$v python3 bench.py --cpu-budget 0 --steps 8 --no-live-pmc 2>/dev/null | python3 tests/probes/bench_fields.py fuse=$v
  ENLSIP_GN_FUSE_NEAR=$v python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py fuse=$v
  ENLSIP_GN_FUSE_NEAR=$v python3 bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py fuse=$v
done; done
