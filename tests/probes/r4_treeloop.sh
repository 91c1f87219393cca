# block-loop kernel of the tree-level updates: parity first, then A/B over the blocks per workgroup (0 = one block per workgroup)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout -k 10 900 python3 -m pytest tests/test_gpu_full_configs.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4t/pytest.log 2>&1 || { tail -30 gpurun_out/r4t/pytest.log; exit 1; }
tail -3 gpurun_out/r4t/pytest.log
for i in 1 2; do for v in 0 4 3 6 2; do
  ENLSIP_GN_TREE_BPW=$v python3 bench.py --cpu-budget 0 --steps 8 --no-live-pmc 2>/dev/null | python3 tests/probes/bench_fields.py bpw=$v
done; done
