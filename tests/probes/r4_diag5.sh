set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
tests/microbench/update_bench_vca4 64 0 0 | grep -i "PAIR host"
for a in "" _vca4 _abl6 _abl7; do for p in 0 6; do UB_EXACT=1 UB_PAIR_ONLY=1 tests/microbench/update_bench$a 384 $p 0 | grep PAIRONLY | sed "s/PAIRONLY/abl[$a]/"; done; done > $O/abl3.txt
cat $O/abl3.txt
