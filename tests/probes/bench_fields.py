"""Print a few fields of bench.py's JSON line (stdin): tag value ms/step latency stages — helper for A/B runs on the GPU box."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(" ".join(sys.argv[1:]), "|", d["config"].get("name"), "value", d["value"], "ms/step", d["ms_per_step"], "lat_ms", d.get("single_problem_latency_ms"),
      "frac", r.get("frac"), "stages", d.get("stage_ms_per_step") or d.get("stage_ms"))
