"""One line of a bench.py result: python tools/print_bench.py <json file> [label]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2] if len(sys.argv) > 2 else "", d["value"], d["ms_per_step"], d.get("p50_latency_ms_b1"), d.get("kernel_ms_per_step", {}).get("lstm"), d.get("roofline", {}).get("frac"))
