set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
B=tests/microbench/update_bench
for b in 48 96 192 384 768; do for p in 0 6; do UB_EXACT=1 UB_PAIR_ONLY=1 $B $b $p 0 | grep PAIRONLY | sed "s/PAIRONLY/batch $b/"; done; done > $O/batch.txt
cat $O/batch.txt
