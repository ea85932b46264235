"""stdin: bench.py's output -> value, ms_per_step and the batches (for quick comparisons).  usage: python bench.py ... | python tools/bench_line.py [label]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["timing"]["batch_ms_per_step"])
