"""Times the SPA stage of the bench workload with parts of spa4_moments switched off ("spa_abl", wrong results)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod

n, block, seed = 430_000, 50_000, 20260
mod = synth.synth_null_model(n, "binary", 0.01, n_cov=3, seed=seed)
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
# carriers of the flagged variants (codes other than the major-allele homozygote, missing included)
sc.scan_2bit_dev(packed.data_ptr(), bpv, block, out.data_ptr(), valid.data_ptr())
sc.sync()
o = out.cpu().numpy(); va = valid.cpu().numpy().astype(bool)
fl = np.where(va & (o[:, 6] <= 0.05))[0]
lutc = torch.tensor([bin(b & 0x55 | (b >> 1) & 0x55).count("1") for b in range(256)], dtype=torch.int32, device=dev)
nnz = []
for j0 in range(0, fl.size, 256):
    rows = packed[torch.from_numpy(fl[j0:j0 + 256]).to(dev)]
    flip = torch.from_numpy((o[fl[j0:j0 + 256], 0] > 0.5)).to(dev)
    rows = torch.where(flip[:, None], rows ^ 0xAA, rows)
    nnz.append(lutc[rows.long()].sum(1).cpu().numpy())
nnz = np.concatenate(nnz)
big = nnz > 16384
print(f"flagged {fl.size}: {big.sum()} with more than 16384 carriers ({nnz[big].sum():.3e} carriers), "
      f"{(~big).sum()} below ({nnz[~big].sum():.3e})", flush=True)
for abl in [int(a) for a in (sys.argv[1:] or ["0", "1", "2", "3", "4", "12"])]:
    sc.set_option("spa_abl", abl)
    for i in range(2):
        sc.scan_2bit_dev(packed.data_ptr(), bpv, block, out.data_ptr(), valid.data_ptr())
    sc.stats_total(reset=True)
    for i in range(5):
        sc.scan_2bit_dev(packed.data_ptr(), bpv, block, out.data_ptr(), valid.data_ptr())
    tot, nc = sc.stats_total(reset=True)
    print(f"abl={abl:3d}  score {tot['ms_score']/nc:.3f} ms  spa {tot['ms_spa']/nc:.3f} ms  slow {tot['n_spa_slow']//nc} of {tot['n_spa']//nc}", flush=True)
sc.close()
