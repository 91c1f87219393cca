cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5prio}; mkdir -p $O
for i in 1 2; do
  for v in 1 0; do
    ENLSIP_GN_BULK_PRIO=$v timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_prio${v}_$i.err | python3 tests/probes/bench_fields.py bulkprio $v >> $O/ab.txt
  done
done
cat $O/ab.txt
