# round 5: tree-level update kernels with the unit's distance in the per-lane offset (no scalar base per unit and column) against the
# previous build (lib/libenlsip_gn_prev.so) — parity subset, then C2 / C4 same-box A/B — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5ta}; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_dispatch_grid.py tests/test_gpu_robustness.py -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for i in 1 2 3; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
  done
done
for i in 1 2; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --config C4 --cpu-budget 0 --steps 5 2> $O/c4_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2> $O/c4s_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
  done
done
cat $O/ab.txt | cut -c1-330
