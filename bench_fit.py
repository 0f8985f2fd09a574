#!/usr/bin/env python3
"""Whole null-model fit on one MI355X (BASELINE config [4]):
``seqFitNullGLMM_SPA`` (AI-REML + PCG over the implicit GRM + variance ratio) on synthetic data.

    python bench_fit.py [--n-samp 430000] [--markers 100000] [--trait binary]

Prints one JSON line.  Not the headline metric (that is bench.py).  The CPU figure beside it is
the oracle's time per implicit-GRM product (one core, scaled from a marker sample) times the
number of products the fit made; the reference's own fit is not runnable here.  The oracle is touched only in that cpu_baseline leg."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-samp", type=int, default=430_000)
    ap.add_argument("--markers", type=int, default=100_000)
    ap.add_argument("--trait", default="binary", choices=["binary", "quantitative"])
    ap.add_argument("--cpu-markers", type=int, default=200, help="markers of the CPU oracle sample; 0 = skip")
    ap.add_argument("--seed", type=int, default=20260)
    args = ap.parse_args()
    import torch
    from saigegds_amd import synth
    from saigegds_amd._lib import GrmOperator, Scanner
    from saigegds_amd.assoc import GenotypeSource
    from saigegds_amd.fitnull import seqFitNullGLMM_SPA
    from saigegds_amd.nullmod import init_nullmod
    if not torch.cuda.is_available():
        raise SystemExit("bench_fit.py needs an MI355X")
    n, m = args.n_samp, args.markers
    dev = torch.device("cuda", 0)
    # synthetic common markers (MAF 0.01..0.5), generated on the device, then brought to the host
    # because the fit's boundary takes an opened genotype file (host memory)
    small = synth.synth_null_model(min(n, 20000), "binary", 0.1, seed=args.seed)
    gen = Scanner(init_nullmod(small, np.arange(min(n, 20000)), float("nan"), 10, 0.1, 0.05, 0.94), 0)
    gen.n = n                       # only the generator of this handle is used
    bpv = (n + 3) // 4
    bpv_dev = ((n + 511) // 512) * 128
    packed = torch.empty((m, bpv_dev), dtype=torch.uint8, device=dev)
    thr = synth.variant_thresholds(0, m, args.seed, log10_maf=(-2.0, -0.3), flip_frac=0.0, miss_rate=1e-3)
    thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    gen.synth_2bit_dev(packed.data_ptr(), bpv_dev, m, 0, args.seed, thr_d.data_ptr())
    gen.sync()
    host = packed[:, :bpv].cpu().numpy()
    del packed
    gen.close()
    torch.cuda.empty_cache()
    # phenotype: covariates as in the reference's example data (x1 ~ N(0,1), x2 ~ Bernoulli(0.5))
    rng = np.random.default_rng(args.seed)
    x1, x2 = rng.standard_normal(n), rng.integers(0, 2, n).astype(np.float64)
    if args.trait == "binary":
        eta = -2.2 + 0.5 * x1 + 0.5 * x2
        y = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    else:
        y = 5 + 0.3 * x1 + 0.3 * x2 + rng.standard_normal(n)
    sid = [f"s{i + 1}" for i in range(n)]
    data = {"sample.id": np.array(sid), "y": y, "x1": x1, "x2": x2}
    src = GenotypeSource(sid, packed=host)

    calls = {"pcg": 0, "pcg_iters": 0, "crossprod": 0, "t_pcg": 0.0, "t_cp": 0.0}

    class Counted(GrmOperator):
        def pcg(self, *a, **k):
            t = time.perf_counter()
            x, it = GrmOperator.pcg(self, *a, **k)
            calls["t_pcg"] += time.perf_counter() - t
            calls["pcg"] += 1
            calls["pcg_iters"] += it
            return x, it

        def crossprod(self, b):
            t = time.perf_counter()
            r = GrmOperator.crossprod(self, b)
            calls["t_cp"] += time.perf_counter() - t
            calls["crossprod"] += 1
            return r

    t0 = time.perf_counter()
    mod = seqFitNullGLMM_SPA("y ~ x1 + x2", data, src, trait_type=args.trait, variant_id=np.arange(1, m + 1),
                             verbose=False, operator_factory=Counted)
    t_fit = time.perf_counter() - t0
    matvecs = calls["pcg_iters"] + calls["pcg"] + calls["crossprod"]     # one product per PCG iteration + start
    line = {"metric": "seqFitNullGLMM_SPA wall time (AI-REML + PCG on the implicit GRM + variance ratio)",
            "trait": args.trait, "n_samples": n, "n_markers": m, "seconds": round(t_fit, 2),
            "tau": [float(mod.tau[0]), float(mod.tau[1])], "converged": bool(mod.converged),
            "var_ratio_mean": float(np.mean(mod.var_ratio)), "n_ratio_markers": int(len(mod.var_ratio)),
            "pcg_solves": calls["pcg"], "pcg_iterations": calls["pcg_iters"], "crossprods": calls["crossprod"],
            "grm_products": matvecs, "seconds_in_pcg": round(calls["t_pcg"], 2),
            "seconds_in_crossprod": round(calls["t_cp"], 2)}
    if args.cpu_markers:
        from oracle import GrmOracle
        mc = min(args.cpu_markers, m)
        orc = GrmOracle(host[:mc], n)
        b = rng.standard_normal(n)
        t0 = time.perf_counter()
        orc.crossprod(b)
        dt = (time.perf_counter() - t0) * (m / mc)
        line["cpu_baseline"] = {"seconds_per_product_scaled": round(dt, 1), "cores": 1, "kind": "port",
                                "sample": f"{mc} of {m} markers, oracle/grm_oracle.c, time scaled by markers",
                                "fit_hours_at_that_rate": round(dt * matvecs / 3600, 1)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
