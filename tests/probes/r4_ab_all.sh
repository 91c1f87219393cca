# working library against enlsip.jl_amd/lib/libenlsip_gn_prev.so on C2 / C3 / C5 (same box, alternating)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for L in libenlsip_gn.so libenlsip_gn_prev.so; do
  ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --cpu-budget 0 --steps 8 --no-live-pmc 2>/dev/null | python3 tests/probes/bench_fields.py $L
  ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py $L
  ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py $L
done; done
