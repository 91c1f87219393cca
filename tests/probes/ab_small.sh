# C5 / C3 with and without the fused small kernel (GPU box)
for f in 1 0; do
  for c in C5 C3; do
    ENLSIP_GN_FUSE_SMALL=$f python bench.py --config $c --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py fused $f
  done
done
