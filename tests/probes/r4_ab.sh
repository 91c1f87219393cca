cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python3 bench.py --cpu-budget 0 --steps 8 2>/dev/null | python3 tests/probes/bench_fields.py $L
  done
done
