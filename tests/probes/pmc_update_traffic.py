"""HBM traffic of the level-0 trailing update from two rocprofv3 PMC passes (diagnostic / evidence tooling, not a test).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    python3 tests/probes/pmc_update_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json] [batch=384]

Counters are collected in separate passes (one counter per run) and corrected as MI355X_MICROARCH.md's HBM / rocprofv3 section
prescribes for gfx950: FETCH_SIZE (KB) tallies 128-byte read requests at 64 bytes -> doubled; WRITE_SIZE (KB) as reported.
The timed step runs as two pipelined halves, so the launches of the LAST 2 x L level-0 far updates are taken and normalised to the
L whole-batch launches that bench.py's roofline leg times (L = 7 pair launches for C2, 14 with ENLSIP_GN_PAIR=0): bytes per launch =
sum over the 2 L half launches / L.  The algorithmic bytes of a pair launch are SURVEY 8d's B_trail of BOTH panels on the far columns
(the pass moves about half of that: ratio ~ 0.55)."""
import csv, hashlib, json, os, sys

KERNEL_PAIR = "k_caqr_update_v4_pair<8>"
KERNEL_PLAIN = "k_caqr_update_v4<8, false>"
KERNEL = KERNEL_PAIR


def last_launches(path, counter, count):
    vals = []
    for r in csv.DictReader(open(path)):
        # the single-problem launches of bench.py's latency leg come last and are skipped by their grid (< 10^5 threads)
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) > 100000:
            vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    return [v for _, v in vals[-count:]]


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 384
    global KERNEL
    m, n, t, PB = 4096, 512, 64, 32
    npan = (n - t + PB - 1) // PB
    paired = any(KERNEL_PAIR in r["Kernel_Name"] for r in csv.DictReader(open(fpath)))
    KERNEL = KERNEL_PAIR if paired else KERNEL_PLAIN
    btrail = lambda k, ncols: batch * 8.0 * (2.0 * (m - k * PB) * ncols + (m - k * PB) * PB + PB * PB)
    per_launch_alg = []
    if paired:
        for k in range(0, npan - 1, 2):
            nfar = (n - t) + 1 - (k + 2) * PB                 # columns beyond the pair, incl. the carried right-hand side
            per_launch_alg.append(btrail(k, nfar) + btrail(k + 1, nfar))
        # (an odd last panel would be a plain launch of the other kernel: not at C2's 14 panels)
    else:
        for k in range(npan):
            per_launch_alg.append(btrail(k, (n - t) - (k + 1) * PB + 1))
    L = len(per_launch_alg)
    nl = 2 * L                            # two pipelined halves
    fetch = last_launches(fpath, "FETCH_SIZE", nl)
    write = last_launches(wpath, "WRITE_SIZE", nl)
    assert len(fetch) == nl and len(write) == nl, (len(fetch), len(write), nl)
    hbm = (2.0 * sum(fetch) + sum(write)) * 1024.0 / L
    alg = sum(per_launch_alg) / L
    rec = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline",
           "kernel": KERNEL, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
           "correction": "gfx950: FETCH_SIZE doubled (128-B requests tallied at 64 B), WRITE_SIZE as reported",
           "hbm_bytes_per_launch_avg": hbm, "algorithmic_bytes_per_launch_avg": alg, "ratio": hbm / alg,
           "config": {"m": m, "n": n, "t": t, "batch": batch},
           # bench.py carries this record as roofline.traffic only while the kernel's source is the one the counters were taken on
           "kernel_source": "enlsip.jl_amd/csrc/gn_kernels_update_v4.hpp",
           "kernel_source_sha256": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                                                   "enlsip.jl_amd", "csrc", "gn_kernels_update_v4.hpp"), "rb").read()).hexdigest()}
    print(json.dumps({k: rec[k] for k in ("hbm_bytes_per_launch_avg", "algorithmic_bytes_per_launch_avg", "ratio")}))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
