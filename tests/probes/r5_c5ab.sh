# round 5: C5 (and C3) same-box A/B of lib/libenlsip_gn.so against lib/libenlsip_gn_prev.so, with the small-problem parity tests first — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5c5}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_dispatch_grid.py tests/test_gpu_defining_properties.py -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
for i in 1 2 3 4; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L timeout -k 10 300 python3 bench.py --config C5 --cpu-budget 0 2> $O/c5_$i.err | python3 tests/probes/bench_fields.py $L >> $O/ab.txt
  done
done
cat $O/ab.txt | cut -c1-300
