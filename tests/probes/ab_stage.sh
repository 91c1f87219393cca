# one-stream stage times of the C2 solve over the batch size (GPU box): which stages scale with the batch
for b in 64 128 192 256 320 384 512; do
  ENLSIP_GN_PIPELINE=0 python bench.py --cpu-budget 0 --batch $b --steps 4 2>/dev/null | python tests/probes/bench_fields.py onestream $b
done
