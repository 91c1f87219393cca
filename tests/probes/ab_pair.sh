# A/B of the panel pairs on the GPU box: small batches of C2 and one C4 row shard
for p in 1 0; do
  for b in 1 4 16 64; do
    ENLSIP_GN_PAIR=$p python bench.py --cpu-budget 0 --no-roofline --batch $b --steps 20 2>/dev/null | python tests/probes/bench_fields.py pair $p batch $b
  done
  ENLSIP_GN_PAIR=$p python bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py pair $p C4-shard-32768
done
