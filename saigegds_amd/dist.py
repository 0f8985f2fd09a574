"""Variant sharding across the GPUs of one node (one process per GPU).

The reference fans the scan out with ``seqParallel(split="by.variant")`` and
concatenates the workers' lists in order (R/assoc_single.r:202-227).  Here each
rank owns one contiguous range of variants (so rank order = variant order), the
model is replicated, and the only exchange is one gather of the result table to
rank 0 -- ``torch.distributed`` over RCCL on GPUs, gloo in CPU tests.  There is
no mid-scan collective.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_range(n_variants: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of rank; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_variants, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_table(out_local, valid_local, n_variants: int, group=None, dst: int = 0):
    """Gather the per-rank [m_r, 8] tables (torch tensors, same device type on
    every rank) into variant order on ``dst``.  Returns (out, valid) on ``dst``
    and (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_variants, r, world) for r in range(world)]
    mmax = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if out_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local table has {out_local.shape[0]} rows, shard has {hi - lo}")
    # pad to the largest shard so a single fixed-size gather suffices
    buf = torch.full((mmax, 9), float("nan"), dtype=torch.float64, device=out_local.device)
    buf[:hi - lo, :8] = out_local
    buf[:hi - lo, 8] = valid_local.to(torch.float64)
    gl = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gl, dst=dst, group=group)
    if rank != dst:
        return None, None
    out = torch.cat([g[:h - l, :8] for g, (l, h) in zip(gl, sizes)], dim=0)
    valid = torch.cat([g[:h - l, 8] for g, (l, h) in zip(gl, sizes)], dim=0).to(torch.uint8)
    return out, valid


def scan_sharded(scanner, packed_local_dev, bpv: int, group=None, n_variants: Optional[int] = None,
                 block: int = 50_000):
    """Scan this rank's HBM-resident shard and gather the table on rank 0.
    ``packed_local_dev``: torch uint8 tensor [m_r, bpv] on this rank's GPU."""
    import torch
    m = packed_local_dev.shape[0]
    out = torch.empty((m, 8), dtype=torch.float64, device=packed_local_dev.device)
    valid = torch.empty((m,), dtype=torch.uint8, device=packed_local_dev.device)
    # blocks of 50 000 variants (.bl_size, R/assoc_single.r:204) on the library's two lanes: the SPA
    # stage of one block runs under the score stage of the next
    scanner.set_option("lanes", 2)
    if packed_local_dev.is_cuda:
        # the library launches on its own non-blocking streams: whatever torch stream produced the
        # shard (an unpack kernel, a non_blocking copy) has to be finished first
        torch.cuda.current_stream(packed_local_dev.device).synchronize()
    for lo in range(0, m, block):
        hi = min(m, lo + block)
        scanner.scan_2bit_dev(packed_local_dev[lo:hi].data_ptr(), bpv, hi - lo, out[lo:hi].data_ptr(),
                              valid[lo:hi].data_ptr())
    scanner.sync()
    if n_variants is None:
        import torch.distributed as dist
        t = torch.tensor([m], dtype=torch.int64, device=out.device)
        dist.all_reduce(t, group=group)
        n_variants = int(t.item())
    return gather_table(out, valid, n_variants, group)
