"""One outer-loop run of a reference problem with the HIP library at the update_working_set seam, trace printed as one JSON
line.  The library is whichever ENLSIP_GN_LIB names (default: the product build), so that two BUILDS of the library can be
compared in one test without loading both into one process (tests/test_reference_problems.py: fast reflector scalars against
the IEEE sqrt / division forms).  usage: python tests/outer_trace_worker.py chained_wood_20|osborne2|hs65"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "enlsip.jl_amd" / "python", ROOT / "tests"):
    sys.path.insert(0, str(p))


def main():
    name = sys.argv[1]
    import ref_problems as rp
    from enlsip_gn import GNSolver
    from hip_backend import HipBackend
    s = GNSolver(device=0)
    try:
        if name == "hs65":
            import hs65
            res = hs65.run(HipBackend(s))
        else:
            res = rp.run({"chained_wood_20": lambda: rp.chained_wood(20), "osborne2": rp.osborne2}[name](), HipBackend(s))
    finally:
        s.close()
    trace = [{k: (float(r[k]) if k in ("alpha", "f") else int(r[k])) for k in ("code", "t", "rankA", "rankJ2", "alpha", "f")} for r in res.trace]
    print(json.dumps({"exit_code": int(res.exit_code), "iterations": int(res.iterations), "f": float(res.f), "trace": trace}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
