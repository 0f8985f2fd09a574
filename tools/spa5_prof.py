import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod
n = int(sys.argv[1]); prev = float(sys.argv[2])
block, seed = 50_000, 20260
mod = synth.synth_null_model(n, "binary", prev, n_cov=3, seed=seed)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
sc = Scanner(sm, device=0)
bpv = sc.row_stride()
dev = torch.device("cuda", 0)
packed = torch.empty((block, bpv), dtype=torch.uint8, device=dev)
out = torch.empty((block, 8), dtype=torch.float64, device=dev)
valid = torch.empty((block,), dtype=torch.uint8, device=dev)
thr = synth.variant_thresholds(0, block, seed)
thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
torch.cuda.synchronize()
sc.synth_2bit_dev(packed.data_ptr(), bpv, block, 0, seed, thr_d.data_ptr())
sc.sync()
for i in range(3):
    sc.scan_2bit_dev(packed.data_ptr(), bpv, block, out.data_ptr(), valid.data_ptr())
    sc.sync()
tot, nc = sc.stats_total(reset=True)
print({k: v / nc for k, v in tot.items()})
sc.close()
