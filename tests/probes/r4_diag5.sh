set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
for a in "" _abl2 _abl3 _abl4 _abl5; do for p in 0 6 10; do UB_EXACT=1 UB_PAIR_ONLY=1 tests/microbench/update_bench$a 384 $p 0 | grep PAIRONLY | sed "s/PAIRONLY/abl[$a]/"; done; done > $O/abl.txt
cat $O/abl.txt
