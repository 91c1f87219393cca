# C4: deep look-ahead A/B + timeline (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5c}; mkdir -p $O
for i in 1 2; do
  for d in 0 1; do
    ENLSIP_GN_LA_DEEP=$d timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_deep${d}_$i.err | python3 tests/probes/bench_fields.py deep $d >> $O/c4_ab.txt
  done
done
cat $O/c4_ab.txt
timeout -k 10 400 bash tests/probes/trace_c4.sh $O/c4trace > $O/trace.log 2>&1; tail -n 75 $O/trace.log
