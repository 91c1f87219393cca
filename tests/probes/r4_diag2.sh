set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
B=tests/microbench/update_bench
for p in 0 2 4 6 8 10; do UB_PAIR_ONLY=1 $B 384 $p 0 | grep PAIRONLY; UB_EXACT=1 UB_PAIR_ONLY=1 $B 384 $p 0 | grep PAIRONLY | sed 's/PAIRONLY/EXACT   /'; done > $O/sweep.txt
cat $O/sweep.txt
python3 bench.py --cpu-budget 0 > $O/c2.json 2> $O/c2.err
python3 -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], json.dumps(d['roofline'].get('all_update_kernels')), d['stage_ms_per_step'])"
