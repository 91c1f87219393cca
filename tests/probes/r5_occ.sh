cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5occ}; mkdir -p $O
for i in 1 2 3; do
  for v in base occ5 occ6; do
    lib=enlsip.jl_amd/lib/libenlsip_gn.so; [ $v != base ] && lib=enlsip.jl_amd/lib/libenlsip_gn_$v.so
    ENLSIP_GN_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config C5 --cpu-budget 0 2> $O/c5_${v}_$i.err | python3 tests/probes/bench_fields.py $v >> $O/ab.txt
  done
done
cat $O/ab.txt
