import sys, time, numpy as np
sys.path.insert(0,"/root/repo")
import torch
from saigegds_amd import synth
from saigegds_amd._lib import GrmOperator, Scanner
from saigegds_amd.nullmod import init_nullmod
n, m = 430000, 100000
dev = torch.device("cuda", 0)
small = synth.synth_null_model(20000, "binary", 0.1, seed=1)
gen = Scanner(init_nullmod(small, np.arange(20000), float("nan"), 10, 0.1, 0.05, 0.94), 0); gen.n = n
bpv = ((n + 511) // 512) * 128
packed = torch.empty((m, bpv), dtype=torch.uint8, device=dev)
thr = synth.variant_thresholds(0, m, 1, log10_maf=(-2.0, -0.3), flip_frac=0.0, miss_rate=1e-3)
thr_d = torch.from_numpy(thr.view(np.int32)).to(dev); torch.cuda.synchronize()
gen.synth_2bit_dev(packed.data_ptr(), bpv, m, 0, 1, thr_d.data_ptr()); gen.sync()
op = GrmOperator(None, n, 0, dev_ptr=packed.data_ptr(), n_markers=m, bytes_per_marker=bpv)
rng = np.random.default_rng(1)
mu = rng.uniform(0.02, 0.4, n); w = mu*(1-mu)
for tau in ([1.0, 0.3], [1.0, 0.0]):
    ts=[]; its=[]
    for k in range(8):
        b = rng.standard_normal(n)
        t=time.perf_counter(); x,it = op.pcg(w, tau, b, 500, 1e-5); ts.append(time.perf_counter()-t); its.append(it)
    print("tau",tau,"ms per solve",[round(x*1e3,1) for x in ts],"iters",its)
t=time.perf_counter()
for k in range(8): op.crossprod(rng.standard_normal(n))
print("crossprod host ms", (time.perf_counter()-t)/8*1e3)
