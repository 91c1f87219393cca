# launch-by-launch list of one one-stream C2 step (name, grid, duration): usage r4_seq.sh OUTDIR [env assignments...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/seq}; mkdir -p $O
export ENLSIP_GN_PIPELINE=0
for kv in "${@:2}"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $O/ks -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline --no-live-pmc $SEQ_ARGS > $O/bench.json 2> $O/ks.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/ks/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "k_constraint" in r["Kernel_Name"]]
sizes = [int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) * int(rows[i]["Grid_Size_Z"]) for i in big]
mx = max(sizes)
starts = [i for i, s in zip(big, sizes) if s == mx]
lo, hi = starts[-2], starts[-1]
with open("$O/seq.txt", "w") as o:
    for r in rows[lo:hi]:
        k = r["Kernel_Name"].split("(")[0].replace("void gn::", "").replace("gn::", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wx = int(r["Workgroup_Size_X"])
        o.write(f"{d:9.1f} us  {k:34s} grid {int(r['Grid_Size_X'])//wx} x {r['Grid_Size_Y']} x {r['Grid_Size_Z']}\n")
PY
rm -rf $O/ks
