import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod
n, m = 430000, 20000
mod = synth.synth_null_model(n, "binary", 0.01, n_cov=3, seed=20260)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
sc = Scanner(sm, 0)
bpv = sc.row_stride()
dev = torch.device("cuda", 0)
pk = torch.empty((m, bpv), dtype=torch.uint8, device=dev)
thr = torch.from_numpy(synth.variant_thresholds(0, m, 20260).view(np.int32)).to(dev); torch.cuda.synchronize()
sc.synth_2bit_dev(pk.data_ptr(), bpv, m, 0, 20260, thr.data_ptr()); sc.sync()
host = pk.cpu().numpy()
host_tight = np.ascontiguousarray(host[:, :(n + 3) // 4])
for name, arr in (("stride=row_stride", host), ("stride=ceil(N/4)", host_tight)):
    for rep in range(2):
        t0 = time.perf_counter(); out, valid = sc.scan_2bit(arr); dt = time.perf_counter() - t0
    print(f"{name}: {m/dt/1e6:.3f} M variants/s, {arr.nbytes/dt/1e9:.1f} GB/s host->result", flush=True)
