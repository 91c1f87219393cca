cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4fz; mkdir -p $O
timeout -k 10 400 python3 tests/probes/fuzz_gpu.py 200 51 > $O/fuzz_single.log 2>&1; echo "fuzz_gpu rc $?"; tail -3 $O/fuzz_single.log
timeout -k 10 300 python3 tests/probes/fuzz_batched.py 50 52 > $O/fuzz_batched.log 2>&1; echo "fuzz_batched rc $?"; tail -3 $O/fuzz_batched.log
timeout -k 10 200 python3 tests/probes/stress_reuse.py 200 43 > $O/stress_reuse.log 2>&1; echo "stress_reuse rc $?"; tail -2 $O/stress_reuse.log
timeout -k 10 120 python3 tests/probes/nan_inputs.py > $O/nan.log 2>&1; echo "nan_inputs rc $?"; tail -2 $O/nan.log
