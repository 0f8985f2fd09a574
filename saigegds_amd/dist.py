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


def gather_buffers(n_variants: int, device, group=None, dst: int = 0):
    """Receive buffers of ``gather_table`` on ``dst`` (None elsewhere): allocate them once, before a timed
    region or when several tables of the same size are gathered."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_rank(group) != dst:
        return None
    mmax = max(hi - lo for lo, hi in (shard_range(n_variants, r, world) for r in range(world)))
    # one allocation per table, the ranks' receive buffers are views of it: with equal shards the gathered table
    # IS that allocation (no concatenation, nothing allocated when the gather runs)
    big_o = torch.zeros((world * mmax, 8), dtype=torch.float64, device=device)
    big_v = torch.zeros((world * mmax,), dtype=torch.uint8, device=device)
    return ([big_o[r * mmax:(r + 1) * mmax] for r in range(world)],
            [big_v[r * mmax:(r + 1) * mmax] for r in range(world)], big_o, big_v)


def gather_table(out_local, valid_local, n_variants: int, group=None, dst: int = 0, recv=None):
    """Gather the per-rank [m_r, 8] tables (torch tensors, same device type on
    every rank) into variant order on ``dst``: the eight doubles of a row as they are, the ``valid`` bytes as
    bytes (80 MB + 1.25 MB per rank at 1.25 M variants).  Returns (out, valid) on ``dst``
    and (None, None) elsewhere.  ``recv``: buffers from ``gather_buffers``."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_variants, r, world) for r in range(world)]
    mmax = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if out_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local table has {out_local.shape[0]} rows, shard has {hi - lo}")
    if hi - lo == mmax:
        so, sv = out_local.contiguous(), valid_local.contiguous()
    else:
        # the short shards are padded to the largest one so that a fixed-size gather suffices
        so = torch.full((mmax, 8), float("nan"), dtype=torch.float64, device=out_local.device)
        sv = torch.zeros((mmax,), dtype=torch.uint8, device=out_local.device)
        so[:hi - lo], sv[:hi - lo] = out_local, valid_local
    if rank == dst and recv is None:
        recv = gather_buffers(n_variants, out_local.device, group, dst)
    dist.gather(so, recv[0] if rank == dst else None, dst=dst, group=group)
    dist.gather(sv, recv[1] if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None, None
    if len(recv) == 4 and all(h - l == mmax for l, h in sizes):
        return recv[2], recv[3]               # equal shards: the receive buffers are the table, in variant order
    out = torch.cat([g[:h - l] for g, (l, h) in zip(recv[0], sizes)], dim=0)
    valid = torch.cat([g[:h - l] for g, (l, h) in zip(recv[1], sizes)], dim=0)
    return out, valid


def scan_shard(scanner, chunks, bpv: int, out, valid, block: int = 50_000):
    """This rank's part of the scan: ``chunks`` -- one torch uint8 tensor [m, bpv] on this rank's GPU or a
    list of them (the shard in the pieces it is held in) -- through ``scan_2bit_dev`` in blocks of ``block``
    variants (.bl_size, R/assoc_single.r:204) on the library's two lanes, results into out / valid at the
    variants' offsets.  Returns with everything queued finished."""
    import torch
    if isinstance(chunks, torch.Tensor):
        chunks = [chunks]
    # the SPA stage of one block runs under the list pass and the score stage of the next
    scanner.set_option("lanes", 2)
    if chunks and chunks[0].is_cuda:
        # the library launches on its own non-blocking streams: whatever torch stream produced the
        # shard (an unpack kernel, a non_blocking copy) has to be finished first
        torch.cuda.current_stream(chunks[0].device).synchronize()
    off = 0
    for ch in chunks:
        m = ch.shape[0]
        for lo in range(0, m, block):
            hi = min(m, lo + block)
            scanner.scan_2bit_dev(ch[lo:hi].data_ptr(), bpv, hi - lo, out[off + lo:off + hi].data_ptr(),
                                  valid[off + lo:off + hi].data_ptr())
        off += m
    scanner.sync()
    return off


def scan_sharded(scanner, packed_local_dev, bpv: int, group=None, n_variants: Optional[int] = None,
                 block: int = 50_000, out=None, valid=None, recv=None):
    """Scan this rank's HBM-resident shard and gather the table on rank 0.
    ``packed_local_dev``: torch uint8 tensor [m_r, bpv] on this rank's GPU, or a list of such tensors.
    ``out`` / ``valid`` / ``recv``: result and receive buffers made beforehand (else allocated here)."""
    import torch
    chunks = [packed_local_dev] if isinstance(packed_local_dev, torch.Tensor) else list(packed_local_dev)
    m = sum(int(c.shape[0]) for c in chunks)
    dev = chunks[0].device
    if out is None:
        out = torch.empty((m, 8), dtype=torch.float64, device=dev)
    if valid is None:
        valid = torch.empty((m,), dtype=torch.uint8, device=dev)
    scan_shard(scanner, chunks, bpv, out, valid, block)
    if n_variants is None:
        import torch.distributed as dist
        t = torch.tensor([m], dtype=torch.int64, device=out.device)
        dist.all_reduce(t, group=group)
        n_variants = int(t.item())
    return gather_table(out, valid, n_variants, group, recv=recv)
