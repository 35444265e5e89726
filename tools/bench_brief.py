"""Prints the few numbers of a bench.py JSON line that are looked at between runs: `python bench.py ... | python tools/bench_brief.py`."""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_blocks"]); print({k:(round(v["avg_launch_ms"],4)) for k,v in d["roofline_iter"]["kernels"].items()}, d["roofline"]["avg_launch_ms"])
