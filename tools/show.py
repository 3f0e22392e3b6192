#!/usr/bin/env python
"""One-line digest of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print(round(d["value"], 1), d.get("repeats", {}).get("frames_per_s"), "latency", round(d.get("single_stream", {}).get("ms_per_frame", 0), 4),
      d.get("stage_ms"), d.get("scene"), "policy", d["config"].get("emission_policy"))
