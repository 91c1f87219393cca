cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
tests/microbench/update_bench_s 384 0 1 > $O/tree.txt 2>&1
cat $O/tree.txt
