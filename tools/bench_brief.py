#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line:  python tools/bench_brief.py gpurun_out/bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], (d.get("cpu_baseline") or {})
print(f"value={d['value']:.0f} {d['unit']}  ms/step={d['ms_per_step']}  kernel_ms={r['avg_launch_ms']}  score_stage_ms={r['stages']['score']['avg_ms']}  frac={r['frac']:.4f}  "
      f"bound={r.get('bound')} frac_of_binding={r.get('frac_of_binding')}  spa_ms={r['stages']['spa']['avg_ms']}  "
      f"whole_step_frac={r.get('whole_step_frac')}  parity={c.get('parity_ok')}")
if d.get("resident_block"):
    print(f"  resident_block: {d['resident_block']['value']:.0f} variants/s  {d['resident_block']['ms_per_step']} ms/step;  block_load {d['block_load']['ms']} ms ({d['block_load']['GBps']} GB/s)")
for k, s in (d.get("secondary") or {}).items():
    if k == "grm":
        print(f"  grm: {s['ms_per_matvec']} ms per mat-vec, {s['roofline']['achieved']} GB/s ({s['roofline']['frac']} of HBM peak), PCG {s['pcg']['iterations']} iterations {s['pcg']['seconds']} s")
        continue
    print(f"  {k}: {s['value']:.0f} variants/s  {s['ms_per_step']} ms/step  kernel {s.get('kernel_ms')}  score {s['score_stage_ms']}  spa {s['spa_stage_ms']}  "
          f"frac {s['frac']}  bound {s['bound']} ({s['frac_of_binding']})")
