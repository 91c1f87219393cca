"""Print the oracle-backend and HIP-backend traces of HS problems side by side (diagnostic)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hs_problems as hp
from oracle import enlsip_outer as eo
from enlsip_gn import GNSolver
from hip_backend import HipBackend

names = sys.argv[1:] or list(hp.PROBLEMS)
for name in names:
    ref = hp.run(name, eo.OracleBackend())
    s = GNSolver(device=0)
    res = hp.run(name, HipBackend(s))
    s.close()
    print(f"== {name}: exit {ref.exit_code} / {res.exit_code}, iterations {ref.iterations} / {res.iterations}")
    for k in range(max(len(ref.trace), len(res.trace))):
        for tag, tr in (("ora", ref.trace), ("hip", res.trace)):
            if k < len(tr):
                a = tr[k]
                print(f"  {k:2d} {tag} " + " ".join(f"{key}={a[key]:.12g}" if isinstance(a[key], float) else f"{key}={a[key]}" for key in a))
