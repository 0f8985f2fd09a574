#!/usr/bin/env python3
"""Benchmark of the single-variant SPA scan (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path -- rows -> result table: the pass that lists the missing genotypes, the score
stage and the SPA stage, through sgx_scan_2bit_dev -- over one block of 50 000 variants (the reference's
seqParallel block size, R/assoc_single.r:204) of synthetic ROW-MAJOR 2-bit genotypes that are already resident in
this GPU's HBM ("sgx_scan_* on HBM-resident packed genotypes").  Every step scans a different block
of rows.  Beside the headline: `resident_block` (sgx_scan_block on blocks loaded once -- what a second, third ...
phenotype over the same genotypes costs per block) and `block_load` (what loading such a block costs).

Workloads; variants shard across ranks, per-GPU work is
fixed (weak scaling); with 8 GPUs and 25 steps the job is BASELINE config [2]:
  c3 (default)  N=430 000, binary trait, prevalence 0.01 (1:99), K=3
  c2            N= 50 000, binary trait, prevalence 0.10 (1:9),  K=3
  c4            N=430 000, quantitative trait (no SPA), K=3
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.3 TB/s measured copy); 256 CUs x 4 SIMDs at 2.4 GHz;
# v_mfma_i32_16x16x64_i8 16 cycles per SIMD, holding the SIMD's vector issue for 8 of them; every other
# vector instruction 4 issue cycles
HBM_PEAK_GBS = 8000.0
CLOCK_HZ = 2.4e9
MFMA_CYCLES, MFMA_ISSUE, VALU_ISSUE = 16, 8, 4
UNPACK_OPS = 9              # vector instructions per (16 variants x 16 samples) dword of score3_kernel (kern_score3.h)
UNPACK_OPS_3 = 15           # ... of its three-plane form (s3_unpack3_op), which also has 2 nbf - 1 MFMAs per dword instead of nbf
BOUND_NAME = {"hbm": "hbm", "mfma": "mfma", "valu_issue": "valu issue"}

WORKLOADS = {
    "c3": dict(n=430_000, trait="binary", prevalence=0.01, desc="configs[2] shard"),
    "c2": dict(n=50_000, trait="binary", prevalence=0.10, desc="configs[1]"),
    "c4": dict(n=430_000, trait="quantitative", prevalence=0.0, desc="configs[3] shard"),
}


def _physical_cores():
    """(physical cores this process may use, CPU model) from /proc/cpuinfo and the affinity mask."""
    model, pairs = "unknown", set()
    try:
        phys = core = None
        allowed = os.sched_getaffinity(0)
        cpu = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                if cpu in allowed:
                    pairs.add((phys, core))
        n = len(pairs) or len(allowed)
    except OSError:
        n = os.cpu_count() or 1
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota) bounds the useful workers
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            q, per = parse(open(path).read())
            if per is None:
                per = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if q != "max" and int(q) > 0:
                n = min(n, max(1, int(q) // int(per)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n), model


class Case:
    """One configuration resident on this rank's GPU: model, scanner, a pool of row-major genotype blocks."""

    def __init__(self, workload, k, block, seed, pool_gb, want_blocks, rank, local, n_override=0, miss_rate=1e-3, n_resident=0):
        import torch
        from saigegds_amd import synth
        from saigegds_amd._lib import Block, Scanner
        from saigegds_amd.nullmod import init_nullmod
        self.torch, self.synth = torch, synth
        self.wl = dict(WORKLOADS[workload])
        if n_override:
            self.wl["n"] = n_override
        self.workload, self.k, self.block, self.seed, self.rank, self.miss_rate = workload, k, block, seed, rank, miss_rate
        self.n = n = self.wl["n"]
        self.dev = torch.device("cuda", local)
        mod = synth.synth_null_model(n, self.wl["trait"], self.wl["prevalence"], n_cov=k, seed=seed)
        self.sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
        self.sc = Scanner(self.sm, device=local)
        self.bpv = self.sc.row_stride()
        self.blk_bytes = block * self.bpv
        self.pool = max(1, min(want_blocks, int(pool_gb * 1e9 // self.blk_bytes)))
        # the resident genotypes: row-major 2-bit rows, the generator's output, as a caller of sgx_scan_2bit_dev holds them
        self.rows = torch.empty((self.pool, block, self.bpv), dtype=torch.uint8, device=self.dev)
        self.out = torch.empty((self.pool, block, 8), dtype=torch.float64, device=self.dev)
        self.valid = torch.empty((self.pool, block), dtype=torch.uint8, device=self.dev)
        t0 = time.time()
        for b in range(self.pool):
            self.generate_rows(b)
        self.t_gen = time.time() - t0
        # a few blocks loaded into the library's resident form (rows + lists of the missing genotypes + carrier lists)
        self.blocks = [Block(n, block, device=local) for _ in range(min(n_resident, self.pool))]
        self.res_bytes = Block.nbytes(n, block)
        self.load_ms = []
        for b, blk in enumerate(self.blocks):
            t = time.perf_counter()
            self.sc.load_block_dev(blk, self.rows[b].data_ptr(), self.bpv, block)
            self.sc.sync()
            self.load_ms.append((time.perf_counter() - t) * 1e3)
        limbs, ngroups = self.sc.score_layout()
        self.limbs = [int(x) for x in limbs]
        self.nbf = (sum(self.limbs) + 1 + 15) // 16 + 1 if ngroups else 0      # B fragments of the score kernel (value + bit-1)

    def generate_rows(self, b):
        """the row-major rows of pool block b (counter-based generator: any block, any time)"""
        torch = self.torch
        first = (self.rank * self.pool + b) * self.block
        thr = self.synth.variant_thresholds(first, self.block, self.seed, miss_rate=self.miss_rate)
        thr_d = torch.from_numpy(thr.view(np.int32)).to(self.dev)
        torch.cuda.synchronize()
        self.sc.synth_2bit_dev(self.rows[b].data_ptr(), self.bpv, self.block, first, self.seed, thr_d.data_ptr())
        self.sc.sync()

    def step(self, i):
        # asynchronous: the library queues the step on one of its streams (alternating with two lanes);
        # its HIP-event stage times are collected after the timed region
        b = i % self.pool
        self.sc.scan_2bit_dev(self.rows[b].data_ptr(), self.bpv, self.block, self.out[b].data_ptr(), self.valid[b].data_ptr())

    def step_resident(self, i):
        b = i % len(self.blocks)
        self.sc.scan_block(self.blocks[b], self.out[b].data_ptr(), self.valid[b].data_ptr())

    def close(self):
        for b in self.blocks:
            b.close()
        self.sc.close()

    def bounds(self, n_cu=256, three_plane=False):
        """(algorithmic bytes per launch, the three rooflines of score3_kernel for this configuration and form)"""
        nfrag, ntile = (self.block + 15) // 16, 2 * ((self.n + 511) // 512)
        row_bytes = math.ceil(self.n / 4)
        alg = self.block * (row_bytes + 64)                       # ceil(N/4) + 64 B per variant
        dwords = nfrag * ntile * 4                                # (16 variants x 16 samples) units
        n_mfma = dwords * (2 * self.nbf - 1 if three_plane else self.nbf)
        n_valu = dwords * (UNPACK_OPS_3 if three_plane else UNPACK_OPS)
        simd_hz = n_cu * 4 * CLOCK_HZ
        return alg, {
            "hbm_ms": round(alg / (HBM_PEAK_GBS * 1e9) * 1e3, 4),
            "mfma_ms": round(n_mfma * MFMA_CYCLES / simd_hz * 1e3, 4),
            "valu_issue_ms": round((n_valu * VALU_ISSUE + n_mfma * MFMA_ISSUE) / simd_hz * 1e3, 4),
            "mfma_per_launch": n_mfma, "valu_per_launch": n_valu, "b_fragments": self.nbf, "form": "three planes" if three_plane else "two planes",
            "rates": "8 TB/s; v_mfma_i32_16x16x64_i8 16 cycles per SIMD (8 of them holding the vector issue port), "
                     "other vector instructions 4 issue cycles; 1024 SIMDs at 2.4 GHz (MI355X_MICROARCH.md)",
        }


def measure(case, steps, warmup, lanes, world=1, dist=None, rehearse=False, resident=False):
    """warmup + K timed steps -> dict(elapsed, stage stats ...).  resident: the steps scan loaded blocks
    (sgx_scan_block) instead of row-major rows.  world > 1: the timed region is the product's own sharded scan
    (saigegds_amd.dist.scan_sharded: the rank's blocks through sgx_scan_2bit_dev on two lanes, then the one
    exchange of the path, the gather of the result table on rank 0)."""
    torch = case.torch
    sc = case.sc
    lanes = max(1, min(lanes, case.pool))             # L steps in flight need L result buffers
    sc.set_option("lanes", lanes)
    for kv in filter(None, os.environ.get("SGX_BENCH_OPTS", "").split(",")):     # experiments: "name=value,..."
        k, v = kv.split("=")
        sc.set_option(k, int(v))
    step = case.step_resident if resident else case.step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    sc.stats_total(reset=True)      # syncs; drops the warm-up steps from the sums
    if world > 1:
        from saigegds_amd import dist as sdist
        # this rank's shard: K blocks of rows (the pool's blocks, round-robin), the table of all of them, and on
        # rank 0 the receive buffers -- everything exists (and is touched) before the clock starts
        chunks = [case.rows[(warmup + i) % case.pool] for i in range(steps)]
        m_r = steps * case.block
        tdev = "cpu" if rehearse else case.dev
        out_r = torch.zeros((m_r, 8), dtype=torch.float64, device=case.dev)
        valid_r = torch.zeros((m_r,), dtype=torch.uint8, device=case.dev)
        recv = sdist.gather_buffers(m_r * world, tdev)
    barrier()
    t0 = time.perf_counter()
    if world == 1:
        for i in range(steps):
            step(warmup + i)
        sc.sync()
    else:
        sdist.scan_shard(sc, chunks, case.bpv, out_r, valid_r, case.block)
        so, sv = (out_r.cpu(), valid_r.cpu()) if rehearse else (out_r, valid_r)
        sdist.gather_table(so, sv, m_r * world, recv=recv)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else case.dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    tot, ncalls = sc.stats_total(reset=True)
    assert ncalls == steps, (ncalls, steps)
    # the stages with nothing running beside them (one lane, a few extra steps outside the timed region)
    iso = None
    if lanes > 1:
        sc.set_option("lanes", 1)
        for i in range(2):                 # (the first steps after the switch still see the other lane's tail)
            step(i)
        sc.stats_total(reset=True)
        per = []
        for i in range(9):
            step(2 + i)
            st = sc.stats()                # (waits for the step: with one lane the next is queued after it anyway)
            per.append((st["ms_score"], st["ms_spa"], st["ms_kernel"], st["ms_lists"]))
        sc.stats_total(reset=True)
        # medians: with the stream idle at submission the events also see the host's gap between two launches,
        # which now and then is hundreds of microseconds
        iso = tuple(float(np.median([p[k] for p in per])) for k in range(4))
        sc.set_option("lanes", lanes)
    return dict(elapsed=elapsed, tot=tot, iso=iso, lanes=lanes)


def main():
    if os.environ.get("SGX_BENCH_WATCHDOG"):      # a run that hangs says where: Python stacks to stderr, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SGX_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--block", type=int, default=50_000, help="variants per step")
    ap.add_argument("--k", type=int, default=3, help="covariates incl. intercept")
    ap.add_argument("--n-samp", type=int, default=0, help="override N (debug)")
    ap.add_argument("--pool-gb", type=float, default=120.0, help="max HBM for resident genotypes")
    ap.add_argument("--missing", type=float, default=1e-3, help="missing-genotype rate of the synthetic rows")
    ap.add_argument("--resident-steps", type=int, default=30, help="steps of the resident-block measurement beside the headline; 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (wall, all cores); 0 = skip")
    ap.add_argument("--seed", type=int, default=20260)
    ap.add_argument("--host-variants", type=int, default=20000,
                    help="variants of one block pushed through the host-buffer entry point (PCIe-inclusive rate); 0 = skip")
    ap.add_argument("--lanes", type=int, default=2, choices=[1, 2, 3, 4],
                    help="library streams per GPU: with 2 the SPA stage of one step runs under the score stage of the next")
    ap.add_argument("--file-variants", type=int, default=12000,
                    help="variants written to a GDS file and scanned from it with seqAssocGLMM_SPA (file -> table rate); 0 = skip")
    ap.add_argument("--grm-markers", type=int, default=100_000, help="markers of the null-model operator's figures in `secondary.grm`; 0 = skip")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1: after the main measurement (rank 0, one GPU) a few steps of K = 13, c2 and c4 into `secondary`")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    # rehearsal on a one-GPU box: SGX_BENCH_REHEARSE=1 puts every rank on device 0 and moves the
    # exchange to gloo (RCCL refuses two ranks on one device); the driver's runs never set it
    rehearse = os.environ.get("SGX_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # one rank per GPU: every rank must sit on a device of its own
        mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device="cpu" if rehearse else dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        devs = [int(x.item()) for x in allv]
        if not rehearse and len(set(devs)) != world:
            raise SystemExit(f"bench.py --gpus {world}: ranks share devices {devs}; launch one rank per GPU (LOCAL_RANK)")
        rccl_ranks = {"backend": "gloo (rehearsal on one device)" if rehearse else "nccl (RCCL)",
                      "world_size": dist.get_world_size(), "devices": devs}

    block, steps, warmup = args.block, args.steps, args.warmup
    do_res = rank == 0 and world == 1 and args.resident_steps > 0
    case = Case(args.workload, args.k, block, args.seed, args.pool_gb, steps + warmup, rank, local, args.n_samp,
                miss_rate=args.missing, n_resident=4 if do_res else 0)
    n, wl, sc, pool = case.n, case.wl, case.sc, case.pool
    r = measure(case, steps, warmup, args.lanes, world, dist, rehearse)
    elapsed, tot, lanes = r["elapsed"], r["tot"], r["lanes"]
    # ---- beside the headline: blocks loaded once, scanned again (a further phenotype over the same genotypes) ----
    resident_block, block_load = None, None
    if do_res and case.blocks:
        rs = args.resident_steps
        rr_ = measure(case, rs, 2, args.lanes, resident=True)
        alg_r = block * (math.ceil(n / 4) + 64)
        resident_block = {
            "value": round(rs * block / rr_["elapsed"], 1), "unit": "variants/s", "ms_per_step": round(rr_["elapsed"] / rs * 1e3, 3),
            "steps": rs, "blocks": len(case.blocks), "entry": "sgx_scan_block on blocks loaded before the clock (sgx_block_load_dev)",
            "kernel_ms": round(rr_["tot"]["ms_kernel"] / rs, 4), "score_stage_ms": round(rr_["tot"]["ms_score"] / rs, 4),
            "spa_stage_ms": round(rr_["tot"]["ms_spa"] / rs, 4),
            "whole_step_frac": round(alg_r * rs / rr_["elapsed"] / 1e9 / HBM_PEAK_GBS, 5),
            "note": "the lists of the missing genotypes and the carrier lists of the rare variants are the block's, made at load; "
                    "what every phenotype after the first costs per block of genotypes"}
        ld = sorted(case.load_ms[1:] or case.load_ms)       # (the first load also pays the first launches)
        ld_ms = ld[len(ld) // 2]
        moved = 2 * block * case.bpv
        block_load = {
            "ms": round(ld_ms, 3), "entry": "sgx_block_load_dev (rows in HBM -> resident block), host-timed incl. the sync",
            "bytes_moved": moved, "GBps": round(moved / (ld_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(moved / (ld_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "block_bytes": case.res_bytes,
            "note": "one pass reads the rows, writes the block's copy and lists the missing genotypes (bytes_moved = rows read + rows "
                    "written); a second, partial pass lists the carriers of the rare variants (their rows once more)"}

    # ---- per-kernel figures (rank 0's launches) ----------------------------
    # HIP events recorded by the library on ITS stream around the list pass (s3_lists_kernel: one read of the rows),
    # the score stage (the sparse pass over the missing genotypes, score3_kernel, s3_reduce_kernel, score3_epilogue)
    # and the SPA stage (spa4_moments / spa4_solve / spa5_kernel).
    ms_score = tot["ms_score"] / steps
    ms_spa = tot["ms_spa"] / steps
    ms_kernel = tot["ms_kernel"] / steps     # HIP events around score3_kernel alone, on the stream it is launched on
    ms_lists = tot["ms_lists"] / steps       # HIP events around the pass that lists the missing genotypes
    n_spa = int(tot["n_spa"])
    n_valid = int(tot["n_valid"])
    nv_tot = steps * block
    row_bytes = math.ceil(n / 4)
    three_plane = tot.get("three_plane", 0) * 2 > steps       # the form the steps took (row-major calls: up to 4 B fragments, or many missing genotypes)
    alg_bytes, bounds = case.bounds(three_plane=three_plane)
    # The kernel that streams the algorithmic bytes is score3_kernel (ONE launch per step, any K).  The SPA
    # stage re-reads only the rows of the flagged variants; it is FP64-issue-bound, not HBM-bound, and is
    # reported as a stage beside it.
    score_kernel = "score3_kernel"
    achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9
    # HBM traffic from the PMC counters: only from a profile of THIS configuration (profiles/README.md)
    traffic, spa_traffic, traffic_src = None, None, None
    pmc_file = os.path.join(ROOT, "profiles", "r04_pmc_stages.json")
    if os.path.exists(pmc_file):
        pm = json.load(open(pmc_file))
        if (pm.get("n_samples") == n and pm.get("variants_per_launch") == block and pm.get("n_covariates") == args.k
                and pm.get("trait") == wl["trait"] and pm.get("workload") == args.workload):
            traffic = pm.get("score_hbm_bytes_per_launch")
            spa_traffic = pm.get("spa_hbm_bytes_per_step")
            traffic_src = ("profiles/r04_pmc_stages.json: rocprofv3 --pmc passes of this configuration and build (separate passes, "
                           "gfx950 FETCH_SIZE correction; counters cannot be read inside a timed run)")
    spa_alg = n_spa / max(1, steps) * row_bytes      # rows of the flagged variants, read once more
    whole_gbs = alg_bytes * steps / elapsed / 1e9
    binding = max(("hbm", "mfma", "valu_issue"), key=lambda b: bounds[b + "_ms"])
    # The roofline that binds: HBM (two-plane form at K = 3: GB/s of algorithmic bytes), or the matrix pipe / vector issue
    # of the contraction (the three-plane form, wide models: int8 MFMA operations per second against the dense peak
    # -- v_mfma_i32_16x16x64_i8 at 16 cycles per SIMD = 2 x the BF16 rate, MI355X_MICROARCH.md)
    mfma_ops = bounds["mfma_per_launch"] * 16 * 16 * 64 * 2
    mfma_peak_tops = 256 * 4 * CLOCK_HZ / MFMA_CYCLES * (16 * 16 * 64 * 2) / 1e12
    mfma_achieved = mfma_ops / (ms_kernel * 1e-3) / 1e12
    by_hbm = binding == "hbm"
    roofline = {
        "bound": "hbm" if by_hbm else "mfma", "binding": BOUND_NAME[binding],
        "kernel": score_kernel, "achieved": round(achieved if by_hbm else mfma_achieved, 2), "peak": HBM_PEAK_GBS if by_hbm else round(mfma_peak_tops, 1),
        "unit": "GB/s" if by_hbm else "TFLOP/s", "frac": round(achieved / HBM_PEAK_GBS if by_hbm else mfma_achieved / mfma_peak_tops, 5),
        "unit_note": None if by_hbm else "int8 multiply-adds of v_mfma_i32_16x16x64_i8 (2 per MAC), exact integer arithmetic; dense peak",
        "hbm": {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5)},
        "traffic": traffic, "traffic_source": traffic_src,
        "bounds": bounds, "frac_of_binding": round(bounds[binding + "_ms"] / ms_kernel, 5),
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms_kernel, 4),
        "launch_ms_note": "HIP events right before and after the launch of score3_kernel on the library's stream, averaged over "
                          "the timed steps (what rocprofv3 --kernel-trace reports for the kernel, profiles/); the kernel is queued "
                          "behind the sparse pass over the missing genotypes, so the events time the kernel and not its wait "
                          "for that pass; with several lanes they still include its wait for CUs held by the other lane's "
                          "kernels; stages.score is the whole score stage (sparse pass, kernel, reduction, epilogue)",
        "whole_step_frac": round(whole_gbs / HBM_PEAK_GBS, 5), "whole_step_gbs": round(whole_gbs, 2),
        "alone": None if r["iso"] is None else {
            "avg_launch_ms": round(r["iso"][2], 4), "achieved": round(alg_bytes / (r["iso"][2] * 1e-3) / 1e9, 2),
            "frac": round(alg_bytes / (r["iso"][2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "score_stage_ms": round(r["iso"][0], 4), "spa_stage_ms": round(r["iso"][1], 4), "lists_ms": round(r["iso"][3], 4),
            "note": "one lane (no SPA stage of the previous step running beside it), medians of 9 steps outside the timed region"},
        "stages": {
            "lists": {"avg_ms": round(ms_lists, 4), "kernel": "s3_lists_t3_kernel", "algorithmic_bytes": block * row_bytes,
                      "frac_of_hbm_peak": round(block * row_bytes / (ms_lists * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms_lists > 0 else None, "bound": "hbm",
                      "note": "the two-plane form's pass over the rows (finds the missing genotypes and gathers their score vectors on the "
                              "spot: the rows are then read twice per step); 0 where the steps took the three-plane form -- up to 4 B "
                              "fragments (K = 3 binary) always, else from ~0.5 % missing genotypes -- which reads the rows once"},
            "score": {"avg_ms": round(ms_score, 4), "launches_per_step": int(tot["score_launches"] // steps),
                      "algorithmic_bytes": alg_bytes, "hbm_bytes": traffic, "bound": BOUND_NAME[binding],
                      "frac_of_hbm_peak": round(alg_bytes / (ms_score * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "spa": {"avg_ms": round(ms_spa, 4), "launches_per_step": int(tot["spa_launches"] // steps),
                    "variants_per_step": n_spa / max(1, steps),
                    "algorithmic_bytes": int(spa_alg), "hbm_bytes": spa_traffic, "bound": "fp64 valu issue",
                    "exact_path_variants_per_step": int(tot["n_spa_slow"]) / max(1, steps),
                    "dense_fallback": int(tot["n_spa_dense"])},
            "note": "stage times are HIP events on the library's streams; with two lanes the stages of "
                    "consecutive steps overlap, so they do not add up to ms_per_step",
        },
    }

    # ---- CPU baseline + parity check (rank 0, N=1 only) ---------------------
    # One oracle worker per physical core over the variants of one timed block (the reference's
    # seqParallel block, R/assoc_single.r:204), bounded by --cpu-seconds of wall time.
    cpu = None
    b0 = warmup % pool
    out, valid, bpv, sm = case.out, case.valid, case.bpv, case.sm
    rows = case.rows[b0]             # the rows of a timed block (resident: the generator is not run again)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import Oracle
        cores, cpu_model = _physical_cores()
        pilot = rows[:64].cpu().numpy()
        orc0 = Oracle(sm)
        t = time.perf_counter()
        orc0.scan_2bit(pilot)
        per = (time.perf_counter() - t) / 64
        ns = int(max(64 * cores, min(block, cores * args.cpu_seconds / max(per, 1e-9))))
        sample = rows[:ns].cpu().numpy()
        bnd = np.linspace(0, ns, cores + 1).astype(int)
        workers = [Oracle(sm) for _ in range(cores)]         # ctypes calls release the GIL

        def job(i):
            return workers[i].scan_2bit(sample[bnd[i]:bnd[i + 1]])
        t = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            parts = list(ex.map(job, range(cores)))
        dt = time.perf_counter() - t
        ref = np.concatenate([p[0] for p in parts])
        ref_valid = np.concatenate([p[1] for p in parts])
        got, got_valid = out[b0, :ns].cpu().numpy(), valid[b0, :ns].cpu().numpy()
        ok = np.array_equal(got_valid, ref_valid)
        v = ref_valid.astype(bool)
        cols = [3, 4, 5] + ([] if sm.quant else [6])
        a, rr = got[v][:, cols], ref[v][:, cols]
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.where(a == rr, 0.0, np.abs(a - rr) / np.abs(rr))
            # beta and SE: relative 1e-10 with an absolute floor of 1e-12 on the z-score
            # beta/SE (the score sum cancels to ~0 under the null; tests/conftest.py)
            zabs = np.abs(ref[v][:, 3]) / ref[v][:, 4]
            tol = np.full_like(rel, 1e-10)
            tol[:, 0] += 1e-12 / np.maximum(zabs, 1e-300)
            tol[:, 1] += 1e-12 / np.maximum(zabs, 1e-300)
        plain_fail = np.any(rel > 1e-10, axis=1)             # rows that needed the floor
        ok = ok and np.array_equal(got[v][:, :3], ref[v][:, :3]) and bool(np.all(rel <= tol))
        if not sm.quant:
            ok = ok and np.array_equal(got[v][:, 7], ref[v][:, 7])
        zmask = zabs > 1e-2      # headline figure on the well-conditioned rows
        # the reference algorithm's own rounding noise: the same oracle with long-double sums,
        # on the rows where GPU and oracle differ most (plus the first rows)
        rowerr = np.zeros(ns)
        rowerr[v] = np.nanmax(rel, axis=1) if rel.size else 0.0
        worst = np.argsort(-rowerr)[:200]
        pick = np.unique(np.concatenate([worst, np.arange(min(ns, 300))]))
        ld, _ = Oracle(sm, long_double=True).scan_2bit(sample[pick])
        vl = ref_valid[pick].astype(bool)
        with np.errstate(invalid="ignore", divide="ignore"):
            e_gpu = np.abs(got[pick][vl][:, cols] / ld[vl][:, cols] - 1)
            e_orc = np.abs(ref[pick][vl][:, cols] / ld[vl][:, cols] - 1)
        cpu = {"value": round(ns / dt, 2), "unit": "variants/s", "cores": cores, "kind": "port",
               "cpu_model": cpu_model, "per_core": round(ns / dt / cores, 2),
               "sample": f"first {ns} variants of timed block {b0} (same data as the GPU), one worker per "
                         f"physical core over contiguous ranges, oracle/saige_oracle.c",
               "seconds": round(dt, 2), "parity_ok": bool(ok), "parity_rows": int(v.sum()),
               "parity_rule": "AF/mac/num/converged bit-exact; pval, p.norm rel 1e-10; beta, SE rel 1e-10 + 1e-12 absolute on z = beta/SE",
               "parity_rows_on_z_floor": int(plain_fail.sum()),
               "parity_max_rel": float(np.nanmax(rel[zmask])) if zmask.any() else 0.0,
               "parity_max_rel_pval": float(np.nanmax(rel[:, 2:])) if rel.size else 0.0,
               "gpu_vs_longdouble_max_rel": float(np.nanmax(e_gpu)),
               "oracle_vs_longdouble_max_rel": float(np.nanmax(e_orc))}

    # ---- PCIe-inclusive rate of the host-buffer entry point (rank 0, N=1 only; never `value`) -------
    host_path = None
    if rank == 0 and world == 1 and args.host_variants > 0:
        from saigegds_amd._lib import PinnedBuffer
        nh = min(block, args.host_variants)
        sc.set_option("lanes", 1)
        with PinnedBuffer((nh, bpv)) as pin:
            pin.array[:] = rows[:nh].cpu().numpy()
            sc.scan_2bit(pin.array[:1000])
            best = float("inf")
            for _ in range(2):
                t = time.perf_counter()
                ho, hv = sc.scan_2bit(pin.array)
                best = min(best, time.perf_counter() - t)
            same = bool(np.array_equal(hv, valid[b0, :nh].cpu().numpy()) and
                        np.array_equal(np.nan_to_num(ho, nan=-7.0), np.nan_to_num(out[b0, :nh].cpu().numpy(), nan=-7.0)))
        pcie = 63.0     # GB/s, PCIe Gen5 x16 (MI355X_MICROARCH.md)
        host_path = {"value": round(nh / best, 1), "unit": "variants/s",
                     "entry": "sgx_scan_2bit (pinned host block in: H2D by chunks, each chunk scanned where it lands; table out)",
                     "variants": nh, "GBps_host_to_result": round(nh * (bpv + 65) / best / 1e9, 2), "pcie_peak_GBps": pcie,
                     "frac_of_pcie": round(nh * (bpv + 65) / best / 1e9 / pcie, 4), "same_table_as_resident_scan": same}

    # ---- file -> table: seqAssocGLMM_SPA on a GDS file (rank 0, N=1 only; never `value`) -----------------
    from_file = None
    if rank == 0 and world == 1 and args.file_variants > 0:
        import tempfile
        from saigegds_amd import assoc as assoc_mod
        from saigegds_amd import synth as synth_mod
        from saigegds_amd.gds_write import write_seqarray_genotypes
        nf = min(block, args.file_variants)
        host_rows = rows[:nf].cpu().numpy()
        mod = synth_mod.synth_null_model(n, wl["trait"], wl["prevalence"], n_cov=args.k, seed=args.seed)
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
            fn = os.path.join(td, "bench_genotypes.gds")
            t = time.perf_counter()
            write_seqarray_genotypes(fn, host_rows[:, :(n + 3) // 4], n, sample_id=mod.sample_id, compress="none")
            t_write = time.perf_counter() - t
            fsize = os.path.getsize(fn)
            old_bl = assoc_mod.BLOCK_SIZE
            assoc_mod.BLOCK_SIZE = max(250, nf // 4)       # a few blocks, so that decoding runs ahead of scanning as it does over a whole file
            try:
                sc.set_option("lanes", 1)
                tm = {}
                t = time.perf_counter()
                ans = assoc_mod.seqAssocGLMM_SPA(fn, mod, mac=10.0, missing=0.1, spa_pval=0.05,
                                                 var_ratio=float(mod.var_ratio[0]), verbose=False, timing=tm)
                t_scan = time.perf_counter() - t
            finally:
                assoc_mod.BLOCK_SIZE = old_bl
        gv = valid[b0, :nf].cpu().numpy().astype(bool)
        go = out[b0, :nf].cpu().numpy()[gv]
        same = bool(len(ans["pval"]) == int(gv.sum()) and np.array_equal(np.asarray(ans["pval"]), go[:, 5])
                    and np.array_equal(np.asarray(ans["beta"]), go[:, 3]) and np.array_equal(np.asarray(ans["AF.alt"]), go[:, 0]))
        from_file = {"value": round(nf / t_scan, 1), "unit": "variants/s",
                     "entry": "seqAssocGLMM_SPA(gdsfile, modobj): GDS file (genotype/data dBit2, stored uncompressed) -> result table, "
                              "blocks decoded one ahead of the block being scanned",
                     "variants": nf, "file_bytes": fsize, "seconds": round(t_scan, 3), "write_seconds": round(t_write, 2),
                     "decode_MBps_of_file_bytes": round(fsize / max(tm.get("decode_s", 0.0), 1e-9) / 1e6, 1),
                     "decode_seconds": round(tm.get("decode_s", 0.0), 3), "scan_seconds": round(tm.get("scan_s", 0.0), 3),
                     "blocks_seconds": round(tm.get("blocks_s", 0.0), 3),
                     "blocks": int(math.ceil(nf / max(250, nf // 4))), "same_table_as_resident_scan": same,
                     "note": "seconds = the whole call (file open, sample matching, model tables on the GPU, the blocks, the result "
                             "table); blocks_seconds = handle + blocks with the decode of block i + 1 (two allele codes per sample "
                             "folded into 2-bit dosage codes by the library's host threads, straight from the mapped file, at "
                             "the rate shown) under the scan of block i; on a file this small the fixed part dominates"}

    config = {
        "workload": f"{args.workload}: {wl['desc']}, N={n} samples x {block} variants/step/GPU, "
                    f"{wl['trait']} trait" + (f" prevalence {wl['prevalence']}" if wl['trait'] == 'binary' else ""),
        "n_samples": n, "variants_per_step_per_gpu": block, "n_covariates": args.k,
        "maf_law": f"10^U(-3.3,-0.3), 10% alt-major, missing {args.missing:g}", "thresholds": "mac=10 missing=0.1 spa.pval=0.05",
        "resident_row_blocks": pool, "sharding": f"variants x{world}", "lanes": lanes,
        "frac_spa": round(n_spa / max(1, nv_tot), 5), "frac_valid": round(n_valid / max(1, nv_tot), 5),
        "gen_seconds": round(case.t_gen, 2),
        "resident_format": "row-major 2-bit rows (4 samples per byte, row stride 128 ceil(N / 512) bytes) in HBM: the input of "
                           "sgx_scan_2bit_dev; nothing is derived from them before the timed region",
        "entry": "sgx_scan_2bit_dev" + (" via saigegds_amd.dist.scan_sharded + gather_table" if world > 1 else ""),
        "row_block_bytes": case.blk_bytes, "score_limbs": case.limbs,
    }
    case.close()
    del case, out, valid, rows, r
    torch.cuda.empty_cache()

    # ---- other configurations, a few steps each (rank 0, N=1 only) -----------------------------------
    secondary = None
    if rank == 0 and world == 1 and args.secondary and args.workload == "c3" and args.k == 3 and not args.n_samp:
        secondary = {}
        for name, w2, k2, miss2 in (("k13", "c3", 13, 1e-3), ("c2", "c2", 3, 1e-3), ("c4", "c4", 3, 1e-3),
                                    ("c3_missing_1e-2", "c3", 3, 1e-2), ("c3_missing_2e-2", "c3", 3, 2e-2)):
            c2 = Case(w2, k2, block, args.seed, min(args.pool_gb, 24.0), 4, rank, local, miss_rate=miss2)
            r2 = measure(c2, 12, 3, args.lanes)
            alg2, b2 = c2.bounds(three_plane=r2["tot"].get("three_plane", 0) * 2 > 12)
            ms2 = r2["tot"]["ms_score"] / 12
            mk2 = r2["tot"]["ms_kernel"] / 12
            bind2 = max(("hbm", "mfma", "valu_issue"), key=lambda b: b2[b + "_ms"])
            secondary[name] = {
                "workload": f"{w2}, K={k2}, N={c2.n}", "value": round(12 * block / r2["elapsed"], 1), "unit": "variants/s",
                "ms_per_step": round(r2["elapsed"] / 12 * 1e3, 3), "steps": 12,
                "kernel_ms": round(mk2, 4), "lists_ms": round(r2["tot"]["ms_lists"] / 12, 4), "score_stage_ms": round(ms2, 4),
                "spa_stage_ms": round(r2["tot"]["ms_spa"] / 12, 4), "missing_rate": miss2,
                "frac": round(alg2 / (mk2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "bound": "hbm" if bind2 == "hbm" else "mfma", "binding": BOUND_NAME[bind2],
                "frac_of_binding": round(b2[bind2 + "_ms"] / mk2, 5),
                "bounds": {k: b2[k] for k in ("hbm_ms", "mfma_ms", "valu_issue_ms", "b_fragments", "form")},
                "whole_step_frac": round(alg2 * 12 / r2["elapsed"] / 1e9 / HBM_PEAK_GBS, 5),
            }
            c2.close()
            del c2
            torch.cuda.empty_cache()
        # BASELINE config [4]: the null-model fit's operator (implicit-GRM mat-vec + one PCG solve) at N = 430 000 x 100 000 markers
        if args.grm_markers > 0:
            from bench_grm import measure_grm
            secondary["grm"] = measure_grm(n, args.grm_markers, 5, 200, args.seed)

    if rank == 0:
        line = {
            "metric": "variants/sec seqAssocGLMM_SPA at N=430K; achieved HBM GB/s vs roofline",
            "value": round(nv_tot * world / elapsed, 1), "unit": "variants/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "dtype_note": "f64 results from int8-MFMA fixed-point sums: results are formed in FP64 from EXACT integer sums: the per-sample score vectors enter as 40/48/56-bit "
                          "fixed point (int8 limbs on v_mfma_i32_16x16x64_i8, int32 accumulation), a quantisation below the rounding "
                          "of the reference's own double sums (cpu_baseline.gpu_vs_longdouble_max_rel against oracle_vs_longdouble_max_rel)",
            "data": "synthetic",
            "config": config, "rccl_ranks": rccl_ranks,
            "roofline": roofline, "resident_block": resident_block, "block_load": block_load,
            "cpu_baseline": cpu, "host_path": host_path, "from_file": from_file, "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
