# default C2 bench twice (stage times from the profiled leg) + single-problem latency; GPU box
for i in 1 2; do python bench.py --cpu-budget 0 2>/dev/null | python tests/probes/bench_fields.py run $i; done
