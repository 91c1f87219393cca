cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
( time python3 bench.py --cpu-budget 0 > $O/c2.json 2> $O/c2.err ) 2> $O/time.txt
python3 -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['frac'], r.get('traffic'), r.get('traffic_source'), r.get('traffic_live_failed'), r.get('real_traffic'), r.get('traffic_counters'))"
cat $O/time.txt
