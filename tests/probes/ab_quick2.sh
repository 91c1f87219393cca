# default C2 bench twice + one-stream stage times + C5 + C3; GPU box
for i in 1 2; do python bench.py --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py run $i; done
ENLSIP_GN_PIPELINE=0 python bench.py --cpu-budget 0 --steps 4 2>/dev/null | python tests/probes/bench_fields.py onestream
python bench.py --config C5 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py c5
python bench.py --config C3 --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py c3
