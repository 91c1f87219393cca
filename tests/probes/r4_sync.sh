cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
for cfg in "57 300 512" "33 300 1024" "129 300 1024"; do for v in 0 2; do timeout -k 10 30 tests/microbench/grid_step_latency $cfg $v; echo "exit $?"; done; done > $O/sync.txt 2>&1
cat $O/sync.txt
