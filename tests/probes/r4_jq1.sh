cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for b in jq1_bench_0 jq1_bench jq1_bench_l1s0 jq1_bench_l0s1; do echo -n "$b: "; tests/microbench/$b 384 | grep k_jq1; done; done
