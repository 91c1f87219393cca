# round 5: soak — every randomised probe at several fresh seeds (GPU box, ~12 minutes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5soak}; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python3 "$@" > $O/$name.log 2>&1; echo "$name rc $? : $(tail -n 1 $O/$name.log)"; }
for s in ${SEEDS:-11 12 13 14}; do
  run fuzz_magnitudes_$s tests/probes/fuzz_magnitudes.py 200 $((1000 + s))
  run fuzz_gpu_$s tests/probes/fuzz_gpu.py 300 $((2000 + s))
  run fuzz_batched_$s tests/probes/fuzz_batched.py 60 $((3000 + s))
  run fuzz_accessors_$s tests/probes/fuzz_accessors.py 80 $((4000 + s))
  run fuzz_tsqr_$s tests/probes/fuzz_tsqr.py 20 $((5000 + s))
done
run stress_reuse tests/probes/stress_reuse.py 300 6001
run stress_pipelined tests/probes/stress_pipelined.py
run stress_threads tests/probes/stress_threads.py
run stress_tsqr tests/probes/stress_tsqr.py
