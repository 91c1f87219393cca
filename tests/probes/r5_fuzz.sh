# round 5: the randomised probes at new seeds, longer than the suite runs them (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5fz}; mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python3 "$@" > $O/$name.log 2>&1; echo "$name rc $?"; tail -n 2 $O/$name.log; }
run fuzz_magnitudes_1 tests/probes/fuzz_magnitudes.py 150 101
run fuzz_magnitudes_2 tests/probes/fuzz_magnitudes.py 150 202
run fuzz_gpu tests/probes/fuzz_gpu.py 400 777
run fuzz_batched tests/probes/fuzz_batched.py 80 778
run fuzz_accessors tests/probes/fuzz_accessors.py 120 779
run fuzz_tsqr tests/probes/fuzz_tsqr.py 30 780
run fuzz_tall_r0 tests/probes/fuzz_tall_r0.py
run stress_reuse tests/probes/stress_reuse.py 200 781
run nan_inputs tests/probes/nan_inputs.py
