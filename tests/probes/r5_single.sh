# round 5: kernel timeline of one C2 solve at batch 1 (tests/probes/trace_single.py) — GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5single}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --config C2 --batch 1 --steps 6 --warmup 3 --cpu-budget 0 --no-roofline --no-live-pmc > $O/bench.json 2> $O/kt.err
python3 tests/probes/trace_single.py $O/kt $O/timeline.txt > $O/summary.txt
cat $O/summary.txt
rm -rf $O/kt
