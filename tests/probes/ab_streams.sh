# C2: the library's two-half split against 2 / 3 / 4 host-driven streams without it; GPU box
python bench.py --cpu-budget 0 --no-roofline 2>/dev/null | python tests/probes/bench_fields.py library-split
for s in 2 3 4; do
  ENLSIP_GN_PIPELINE=0 python bench.py --cpu-budget 0 --no-roofline --streams $s 2>/dev/null | python tests/probes/bench_fields.py streams $s
done
ENLSIP_GN_PIPELINE=0 python bench.py --cpu-budget 0 --no-roofline --streams 3 --batch 576 2>/dev/null | python tests/probes/bench_fields.py streams 3 batch 576
python bench.py --cpu-budget 0 --no-roofline --batch 512 2>/dev/null | python tests/probes/bench_fields.py library-split batch 512
