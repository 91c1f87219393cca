# round 4, first diagnosis call: (1) the pair's far update over anchors / partial tiles in the harness,
# (2) kernel trace of ONE C2 solve (single-problem latency by kernel), (3) baseline bench line of the box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
B=tests/microbench/update_bench
for p in 0 2 4 6 8 10; do UB_PAIR_ONLY=1 $B 384 $p 0 | grep PAIRONLY; done > $O/sweep_m4096.txt
# full tiles at an odd pair's anchor (m + 64: 128 blocks from row 64), partial last tile at anchor 0 (m - 64)
UB_M=4160 UB_PAIR_ONLY=1 $B 384 2 0 | grep PAIRONLY >> $O/sweep_other.txt
UB_M=4032 UB_PAIR_ONLY=1 $B 384 0 0 | grep PAIRONLY >> $O/sweep_other.txt
UB_M=4224 UB_PAIR_ONLY=1 $B 384 6 0 | grep PAIRONLY >> $O/sweep_other.txt
UB_M=3968 UB_PAIR_ONLY=1 $B 384 0 0 | grep PAIRONLY >> $O/sweep_other.txt
cat $O/sweep_m4096.txt $O/sweep_other.txt
python3 bench.py --cpu-budget 0 > $O/c2.json 2> $O/c2.err
python3 tests/probes/bench_fields.py base < $O/c2.json || true
export ENLSIP_GN_PIPELINE=0
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --batch 1 --steps 3 --warmup 1 --cpu-budget 0 --no-roofline > $O/single.json 2> $O/kt.err
cp $O/kt/*/*kernel_trace.csv $O/single_kernel_trace.csv; rm -rf $O/kt
echo done
