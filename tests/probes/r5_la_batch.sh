# round 5: the look-ahead sweep (a pair's far update on a second stream beside the next pair's chain) forced on the C2 BATCH — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5lab}; mkdir -p $O
for i in 1 2; do
  for p in 1 0; do
    for x in 0 1; do
      if [ $x = 1 ]; then export ENLSIP_GN_LOOKAHEAD=1; else unset ENLSIP_GN_LOOKAHEAD; fi
      ENLSIP_GN_PIPELINE=$p timeout -k 10 300 python3 bench.py --cpu-budget 0 --no-live-pmc --steps 10 2> $O/c2_p${p}_x${x}_$i.err | python3 tests/probes/bench_fields.py "pipeline $p lookahead" $x >> $O/ab.txt
    done
  done
done
cat $O/ab.txt
