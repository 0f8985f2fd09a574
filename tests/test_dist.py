"""world_size-2 gloo test of the variant sharding + result gather (CPU).
The per-rank compute is stood in for by the CPU oracle (test infrastructure);
what is under test is saigegds_amd/dist.py: contiguous shards, ordered gather."""
import os
import socket
import sys

import numpy as np
import pytest


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from conftest import scan_model
    from oracle import Oracle
    from saigegds_amd.dist import gather_table, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = np.load(os.path.join(root, "tests", "golden", "grm1k_10k_snp.npz"))
        m = 1001                                   # odd: shards of 501 and 500
        lo, hi = shard_range(m, rank, world)
        out, valid = Oracle(scan_model("saige_model.npz", mac=40)).scan_2bit(z["packed"][lo:hi])
        o, v = gather_table(torch.from_numpy(out), torch.from_numpy(valid), m)
        if rank == 0:
            q.put((o.numpy(), v.numpy()))
        else:
            assert o is None and v is None
    finally:
        dist.destroy_process_group()


def _worker_sharded(rank, world, port, q):
    """scan_sharded's own block loop on every rank, the GPU scanner stood in for by the oracle."""
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from conftest import scan_model
    from oracle import Oracle
    from saigegds_amd.dist import scan_sharded, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = np.load(os.path.join(root, "tests", "golden", "grm1k_10k_snp.npz"))
        m = 1001
        lo, hi = shard_range(m, rank, world)
        shard = torch.from_numpy(np.ascontiguousarray(z["packed"][lo:hi]))
        orc = Oracle(scan_model("saige_model.npz", mac=40))
        calls = []

        class HostScanner:                 # the scan_2bit_dev / sync / set_option surface of Scanner
            def set_option(self, name, value):
                calls.append((name, value))

            def scan_2bit_dev(self, p_packed, bpv, m_blk, p_out, p_valid):
                import ctypes
                pk = np.frombuffer((ctypes.c_uint8 * (m_blk * bpv)).from_address(p_packed), dtype=np.uint8).reshape(m_blk, bpv)
                o, v = orc.scan_2bit(pk)
                np.frombuffer((ctypes.c_double * (m_blk * 8)).from_address(p_out), dtype=np.float64)[:] = o.ravel()
                np.frombuffer((ctypes.c_uint8 * m_blk).from_address(p_valid), dtype=np.uint8)[:] = v
                calls.append(("scan", m_blk))

            def sync(self):
                calls.append(("sync", 0))

        o, v = scan_sharded(HostScanner(), shard, shard.shape[1], block=200)      # 3 blocks, ragged tail
        assert [c for c in calls if c[0] == "scan"] == [("scan", 200), ("scan", 200), ("scan", hi - lo - 400)]
        assert ("lanes", 2) in calls and calls[-1][0] == "sync"
        if rank == 0:
            q.put((o.numpy(), v.numpy()))
        else:
            assert o is None and v is None
    finally:
        dist.destroy_process_group()


def _run_two_ranks(target):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    o, v = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return o, v


def test_two_rank_scan_sharded_block_loop():
    from conftest import scan_model, GOLDEN
    from oracle import Oracle
    o, v = _run_two_ranks(_worker_sharded)
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    ref, ref_valid = Oracle(scan_model("saige_model.npz", mac=40)).scan_2bit(z["packed"][:1001])
    assert np.array_equal(v, ref_valid) and np.array_equal(o, ref, equal_nan=True)


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    from conftest import scan_model, GOLDEN
    from oracle import Oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    o, v = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    ref, ref_valid = Oracle(scan_model("saige_model.npz", mac=40)).scan_2bit(z["packed"][:1001])
    assert np.array_equal(v, ref_valid) and (v == 0).any()
    assert np.array_equal(o, ref, equal_nan=True)
