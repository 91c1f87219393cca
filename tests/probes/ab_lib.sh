# same-box A/B of two builds: lib/libenlsip_gn.so against lib/libenlsip_gn_prev.so (single-problem latency + C2 throughput); GPU box
for i in 1 2 3; do
  for L in libenlsip_gn.so libenlsip_gn_prev.so; do
    ENLSIP_GN_LIB=$PWD/enlsip.jl_amd/lib/$L python bench.py --cpu-budget 0 --no-roofline --steps 6 2>/dev/null | python tests/probes/bench_fields.py $L
  done
done
