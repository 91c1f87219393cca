set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/measure}; mkdir -p $O
echo "[1] bench default"; python3 bench.py > $O/c2.json 2> $O/c2.err
echo "[2] kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 > $O/c2_under_rocprof.json 2> $O/ks.err
cp $O/ks/*/*kernel_stats.csv $O/c2_kernel_stats_bench_steps3.csv
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/ks/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_caqr_update_v4_pair<8>" in r["Kernel_Name"]]
by = {}
for r in rows:
    nb = int(r["Grid_Size_Z"]); by.setdefault(nb, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
json.dump({str(k): {"launches": len(v), "avg_us": sum(v) / len(v)} for k, v in sorted(by.items())}, open("$O/update_launches_by_size.json", "w"), indent=1)
print(open("$O/update_launches_by_size.json").read())
PY
rm -rf $O/ks
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  echo "[3] pmc $c"; rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline > /dev/null 2> $O/pmc_$c.err
  cp $O/pmc_$c/*/*counter_collection.csv $O/pmc_$c.csv; rm -rf $O/pmc_$c
done
python3 tests/probes/pmc_update_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/update_traffic_pmc.json
python3 tests/probes/pmc_mfma_util.py $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv $O/pmc_GRBM_GUI_ACTIVE.csv $O/mfma_util_pmc.json
echo "[4] other configs"
python3 bench.py --config C3 > $O/c3.json 2> $O/c3.err
python3 bench.py --config C5 > $O/c5.json 2> $O/c5.err
python3 bench.py --config C4 --steps 5 > $O/c4.json 2> $O/c4.err
python3 bench.py --config C4 --steps 5 --rows 32768 --cpu-budget 0 > $O/c4_shard32768.json 2> $O/c4s.err || true
echo done
