# C3 / C5: working library against two archived builds (enlsip.jl_amd/lib/libenlsip_gn_half.so, libenlsip_gn_prev.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for L in libenlsip_gn.so libenlsip_gn_half.so libenlsip_gn_prev.so; do for c in C3 C5; do
  ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config $c --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py $L
done; done; done
