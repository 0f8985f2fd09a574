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
