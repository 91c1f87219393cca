set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
B=tests/microbench/update_bench
$B 64 0 0 | grep -i "check" > $O/check.txt; cat $O/check.txt
for p in 0 2 4 6 8 10 12; do UB_EXACT=1 UB_PAIR_ONLY=1 $B 384 $p 0 | grep PAIRONLY | sed 's/PAIRONLY/EXACT   /'; done > $O/sweep.txt
cat $O/sweep.txt
python3 bench.py --cpu-budget 0 > $O/c2.json 2> $O/c2.err
python3 -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(d['value'], d['single_problem_latency_ms'], d['roofline']['frac'], [x['frac'] for x in d['roofline']['per_launch']], json.dumps(d['roofline'].get('all_update_kernels')), d['stage_ms_per_step'])"
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
