# C4 on one GPU with and without the look-ahead sweep; GPU box
for l in 1 0; do
  ENLSIP_GN_LOOKAHEAD=$l python bench.py --config C4 --steps 5 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py lookahead $l
done
ENLSIP_GN_PAIR=1 python bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py shard pairs+lookahead
ENLSIP_GN_PAIR=1 ENLSIP_GN_LOOKAHEAD=0 python bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py shard pairs
python bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py shard auto
