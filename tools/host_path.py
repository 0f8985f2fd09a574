"""PCIe-inclusive throughput of sgx_scan_2bit on host buffers (N = 430 000): pageable and pinned source."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from saigegds_amd import synth  # noqa: E402
from saigegds_amd._lib import PinnedBuffer, Scanner  # noqa: E402
from saigegds_amd.nullmod import init_nullmod  # noqa: E402

n, m, seed = 430_000, 50_000, 20260
mod = synth.synth_null_model(n, "binary", 0.01, seed=seed)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
sc = Scanner(sm, device=0)
bpv = sc.row_stride()
dev = torch.device("cuda", 0)
pk = torch.empty((m, bpv), dtype=torch.uint8, device=dev)
thr = torch.from_numpy(synth.variant_thresholds(0, m, seed).view(np.int32)).to(dev)
torch.cuda.synchronize()
sc.synth_2bit_dev(pk.data_ptr(), bpv, m, 0, seed, thr.data_ptr())
sc.sync()
host = pk.cpu().numpy()
with PinnedBuffer((m, bpv)) as pin:
    pin.array[:] = host
    for name, src in (("pageable", host), ("pinned", pin.array)):
        sc.scan_2bit(src[:2000])
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            out, valid = sc.scan_2bit(src)
            best = min(best, time.perf_counter() - t)
        print(f"{name:9s}: {m / best / 1e6:.3f} M variants/s, {m * (bpv + 65) / best / 1e9:.1f} GB/s host->result, "
              f"valid {int(valid.sum())}", flush=True)
sc.close()
