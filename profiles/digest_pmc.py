#!/usr/bin/env python3
"""HBM traffic per kernel from the rocprofv3 --pmc passes of a bench.py run, with the corrections of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and counts 128-B requests of a wide coalesced
stream at 64 B, so the bytes of a kernel's reads are 2 x FETCH_SIZE where its requests are not 32-B
ones (TCC_EA0_RDREQ_32B_sum = 0); WRITE_SIZE (KB) is exact.

    python profiles/digest_pmc.py <read-pass dir> <write-pass dir> <steps in the run> <out.json> key=value ...
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                n[k] += 1
    return acc, n


rd, nr = load(sys.argv[1])
wr, nw = load(sys.argv[2])
steps = int(sys.argv[3])
meta = dict(kv.split("=", 1) for kv in sys.argv[5:])
for k in ("n_samples", "variants_per_launch", "n_covariates"):
    if k in meta:
        meta[k] = int(meta[k])
kern = {}
for k in sorted(set(rd) | set(wr)):
    if "synth" in k or "rocclr" in k or k.startswith("s3_ingest"):     # generator, runtime copies, block loading: not part of a step
        continue
    fetch_kb, r32 = rd[k].get("FETCH_SIZE", 0.0), rd[k].get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rdreq = rd[k].get("TCC_EA0_RDREQ_sum", 0.0)
    wide = rdreq > 0 and r32 < 0.05 * rdreq
    read_b = fetch_kb * 1024 * (2 if wide else 1)
    write_b = wr[k].get("WRITE_SIZE", 0.0) * 1024
    kern[k] = {"launches": nr[k], "read_bytes_per_step": read_b / steps, "write_bytes_per_step": write_b / steps,
               "fetch_doubled": bool(wide)}
score = [v for k, v in kern.items() if k.startswith("score3_kernel") or k.startswith("score_mfma_kernel")]
spa = [v for k, v in kern.items() if k.startswith("spa")]
lists = [v for k, v in kern.items() if k.startswith("s3_lists_t3_kernel") or k.startswith("s3_lists_kernel")]
out = dict(meta)
out["steps_profiled"] = steps
out["score_hbm_bytes_per_launch"] = int(sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in score))
out["spa_hbm_bytes_per_step"] = int(sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in spa))
out["lists_hbm_bytes_per_launch"] = int(sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in lists))
out["all_kernels_hbm_bytes_per_step"] = int(sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in kern.values()))
out["kernels"] = kern
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
