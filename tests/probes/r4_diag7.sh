set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
for a in "" _vb3 _ls0 _vb3ls0; do tests/microbench/update_bench$a 64 0 0 | grep -i "PAIR host"; for p in 0 6 10; do UB_EXACT=1 UB_PAIR_ONLY=1 tests/microbench/update_bench$a 384 $p 0 | grep PAIRONLY | sed "s/PAIRONLY/var[$a]/"; done; done > $O/var.txt
cat $O/var.txt
