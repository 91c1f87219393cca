# per-kernel times of the one-stream C2 step (no pipelining: kernel durations are not stretched by a second stream)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/onestream}; mkdir -p $O
export ENLSIP_GN_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 4 --warmup 1 --cpu-budget 0 --no-roofline ${@:2} > $O/bench.json 2> $O/ks.err
cp $O/ks/*/*kernel_stats.csv $O/kernel_stats.csv
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/ks/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full-batch solve: from the last k_constraint launch with the largest grid to the end
agg = collections.OrderedDict()
big = [i for i, r in enumerate(rows) if "k_constraint" in r["Kernel_Name"]]
sizes = [int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) * int(rows[i]["Grid_Size_Z"]) for i in big]
mx = max(sizes)
starts = [i for i, s in zip(big, sizes) if s == mx]
lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    k = r["Kernel_Name"].split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
tot = (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3
with open("$O/one_step_by_kernel.txt", "w") as o:
    o.write(f"one step, wall {tot:.1f} us, kernels {hi - lo}\n")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{d:10.1f} us {c:5d} x {k}\n")
print(open("$O/one_step_by_kernel.txt").read())
PY
rm -rf $O/ks
