cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C4 $L
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C4shard $L
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C3 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C3 $L
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --config C5 --cpu-budget 0 2>/dev/null | python3 tests/probes/bench_fields.py C5 $L
  done
done
