#!/usr/bin/env python3
"""Benchmark of the null-model operator (BASELINE config [4]):
implicit-GRM mat-vec out = G'(G b)/M on 2-bit genotypes and one PCG solve.

    python bench_grm.py [--n-samp 430000] [--markers 100000] [--reps 5]

Prints one JSON line.  Not the headline metric (that is bench.py).  The oracle is touched only in the
cpu_baseline leg (timing + parity check), as in bench.py."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def measure_grm(n=430_000, m=100_000, reps=5, cpu_markers=300, seed=20260):
    """The operator's figures as a dict (bench.py folds it into `secondary.grm`)."""
    import torch
    from saigegds_amd import synth
    from saigegds_amd._lib import GrmOperator, Scanner
    from saigegds_amd.nullmod import init_nullmod
    if not torch.cuda.is_available():
        raise SystemExit("bench_grm.py needs an MI355X")
    dev = torch.device("cuda", 0)
    # synthetic common markers (MAF 0.01..0.5), generated on the device
    mod = synth.synth_null_model(min(n, 20000), "binary", 0.1, seed=seed)
    sm = init_nullmod(mod, np.arange(min(n, 20000)), float("nan"), 10, 0.1, 0.05, 0.94)
    gen = Scanner(sm, 0)
    gen.n = n                       # only the generator of this handle is used
    bpv = ((n + 255) // 256) * 64
    packed = torch.empty((m, bpv), dtype=torch.uint8, device=dev)
    thr = synth.variant_thresholds(0, m, seed, log10_maf=(-2.0, -0.3), flip_frac=0.0, miss_rate=1e-3)
    thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    gen.synth_2bit_dev(packed.data_ptr(), bpv, m, 0, seed, thr_d.data_ptr())
    gen.sync()
    t0 = time.perf_counter()
    op = GrmOperator(None, n, 0, dev_ptr=packed.data_ptr(), n_markers=m, bytes_per_marker=bpv)
    t_init = time.perf_counter() - t0
    rng = np.random.default_rng(1)
    b = torch.from_numpy(rng.standard_normal(n)).to(dev)
    out = torch.empty_like(b)
    torch.cuda.synchronize()
    op.crossprod_dev(b.data_ptr(), out.data_ptr()); op.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.crossprod_dev(b.data_ptr(), out.data_ptr())
    op.sync()
    t_mv = (time.perf_counter() - t0) / reps
    mu = rng.uniform(0.02, 0.4, n)
    t0 = time.perf_counter()
    x, iters = op.pcg(mu * (1 - mu), [1.0, 0.3], b.cpu().numpy(), 500, 1e-5)
    t_pcg = time.perf_counter() - t0
    packed_bytes = m * ((n + 3) // 4)
    gbs = 2 * packed_bytes / t_mv / 1e9
    line = {"metric": "implicit-GRM mat-vec G'(Gb)/M on 2-bit genotypes", "n_samples": n, "n_markers": m,
            "ms_per_matvec": round(t_mv * 1e3, 3),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                         "algorithmic_bytes_per_matvec": 2 * packed_bytes,
                         "note": "two sweeps per product (over the markers, then over the samples), each streaming the packed "
                                 "matrix once; wall time of back-to-back products between two syncs"},
            "init_seconds": round(t_init, 2),
            "pcg": {"iterations": iters, "seconds": round(t_pcg, 3), "tol": 1e-5, "tau": [1.0, 0.3],
                    "reference": "PCG_diag_sigma, src/saige_fitnull.cpp:581-614"}}
    if cpu_markers:
        from oracle import GrmOracle
        mc = min(cpu_markers, m)
        orc = GrmOracle(packed[:mc].cpu().numpy(), n)
        bh = b.cpu().numpy()
        t0 = time.perf_counter()
        ref = orc.crossprod(bh)
        dt = time.perf_counter() - t0
        sub = GrmOperator(packed[:mc].cpu().numpy(), n)
        got = sub.crossprod(bh)
        sub.close()
        line["cpu_baseline"] = {"ms_per_matvec_scaled": round(dt * m / mc * 1e3, 1), "cores": 1, "kind": "port",
                                "sample": f"{mc} of {m} markers, oracle/grm_oracle.c (get_crossprod_b_grm, src/saige_fitnull.cpp:435-536), time scaled by markers",
                                "parity_max_rel": float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))}
    op.close()
    gen.n = min(n, 20000)
    gen.close()
    del packed, b, out
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-samp", type=int, default=430_000)
    ap.add_argument("--markers", type=int, default=100_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-markers", type=int, default=300, help="markers of the CPU oracle sample; 0 = skip")
    ap.add_argument("--seed", type=int, default=20260)
    args = ap.parse_args()
    print(json.dumps(measure_grm(args.n_samp, args.markers, args.reps, args.cpu_markers, args.seed)), flush=True)


if __name__ == "__main__":
    main()
