"""Reads bench.py's JSON line(s) from stdin and prints the few numbers an A/B looks at."""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("ms_per_step", d["ms_per_step"], "parity", d["parity_vs_oracle"]["ok"], d["kernel_us_per_frame"],
              "latency", (d.get("latency_us") or {}).get("median"), "unfused", (d.get("per_frame_protocol") or {}).get("unfused_ms_per_step"))
