#!/usr/bin/env python3
"""usage: scripts/show_bench.py bench_output.json : the headline figures of a bench.py line (last line of the file)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["steps"], "steps:", d["ms_per_step"], "ms/step,", d["value"], d["unit"], "| fwd", d["roofline"]["fwd_ms"], "adj", d["roofline"]["adj_ms"], "frac", d["roofline"]["frac"])
print({k: v for k, v in d["other_launches_ms"].items() if not isinstance(v, dict)})
