# solves/s of the default C2 bench over the batch size (GPU box)
for b in 256 384 448 512 640 768 1024; do
  python bench.py --cpu-budget 0 --no-roofline --batch $b --steps 6 2>/dev/null | python tests/probes/bench_fields.py batch $b
done
