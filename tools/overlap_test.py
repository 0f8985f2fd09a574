import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod
n, block, nblk = 430000, 50000, 8
mod = synth.synth_null_model(n, "binary", 0.01, n_cov=3, seed=20260)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
dev = torch.device("cuda", 0)
for nh in (1, 2, 3):
    scs = [Scanner(sm, device=0) for _ in range(nh)]
    bpv = scs[0].row_stride()
    packed = torch.empty((nblk, block, bpv), dtype=torch.uint8, device=dev)
    out = torch.empty((nblk, block, 8), dtype=torch.float64, device=dev)
    valid = torch.empty((nblk, block), dtype=torch.uint8, device=dev)
    for b in range(nblk):
        thr = torch.from_numpy(synth.variant_thresholds(b * block, block, 20260).view(np.int32)).to(dev)
        torch.cuda.synchronize()
        scs[0].synth_2bit_dev(packed[b].data_ptr(), bpv, block, b * block, 20260, thr.data_ptr())
        scs[0].sync()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(nblk):
            scs[b % nh].scan_2bit_dev(packed[b].data_ptr(), bpv, block, out[b].data_ptr(), valid[b].data_ptr())
        for sc in scs:
            sc.sync()
        dt = time.perf_counter() - t0
    print(f"handles={nh}: {dt / nblk * 1e3:.3f} ms per block, {nblk * block / dt / 1e6:.2f} M variants/s", flush=True)
    ref = out.clone() if nh == 1 else ref
    if nh > 1:
        print("  identical to single-stream results:", bool(torch.equal(torch.nan_to_num(out), torch.nan_to_num(ref))), flush=True)
    for sc in scs:
        sc.close()
    del packed, out, valid
