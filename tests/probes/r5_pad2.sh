cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5p}; mkdir -p $O
for i in 1 2; do
  for pad in 0 6144; do
    ENLSIP_GN_BULK_PAD=$pad timeout -k 10 300 python3 bench.py --config C4 --steps 5 --cpu-budget 0 2> $O/c4_pad${pad}_$i.err | python3 tests/probes/bench_fields.py pad $pad >> $O/c4_ab.txt
  done
done
cat $O/c4_ab.txt
