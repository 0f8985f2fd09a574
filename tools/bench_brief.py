#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line:  python tools/bench_brief.py gpurun_out/bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], (d.get("cpu_baseline") or {})
print(f"value={d['value']:.0f} {d['unit']}  ms/step={d['ms_per_step']}  score_ms={r['avg_launch_ms']}  frac={r['frac']:.4f}  "
      f"GB/s={r['achieved']:.0f}  spa_ms={r['stages']['spa']['avg_ms']}  parity={c.get('parity_ok')}")
