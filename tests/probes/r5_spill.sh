# round 5: lane geometry re-formed after the step loop of the panel kernels (no spill slot reloaded per step) against the previous
# build (lib/libenlsip_gn_prev.so) — parity subset, then C5 / C3 / C2 same-box A/B — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5sp}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_dispatch_grid.py -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
for i in 1 2 3; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --config C5 --cpu-budget 0 2> $O/c5_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --config C3 --cpu-budget 0 2> $O/c3_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
  done
done
sort -k3,3 -s $O/ab.txt | cut -c1-330
