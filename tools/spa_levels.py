import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod
n=430000; block=50000
mod = synth.synth_null_model(n, "binary", 0.01, n_cov=3, seed=20260)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
dev=torch.device("cuda",0)
sc = Scanner(sm, device=0); bpv=sc.row_stride()
packed = torch.empty((2, block, bpv), dtype=torch.uint8, device=dev)
out = torch.empty((2, block, 8), dtype=torch.float64, device=dev); valid = torch.empty((2, block), dtype=torch.uint8, device=dev)
for b in range(2):
    thr = synth.variant_thresholds(b*block, block, 20260); thr_d = torch.from_numpy(thr.view(np.int32)).to(dev); torch.cuda.synchronize()
    sc.synth_2bit_dev(packed[b].data_ptr(), bpv, block, b*block, 20260, thr_d.data_ptr()); sc.sync()
ref=None
for lv in (12, 8, 6, 5, 4, 3):
    sc.set_option("spa_levels", lv)
    for i in range(2): sc.scan_2bit_dev(packed[i%2].data_ptr(), bpv, block, out[i%2].data_ptr(), valid[i%2].data_ptr()); sc.stats()
    t=time.perf_counter(); sp=0; slow=0
    for i in range(8):
        sc.scan_2bit_dev(packed[i%2].data_ptr(), bpv, block, out[i%2].data_ptr(), valid[i%2].data_ptr()); st=sc.stats(); sp+=st["ms_spa"]; slow+=st["n_spa_slow"]
    dt=(time.perf_counter()-t)/8
    o=out.cpu().numpy().copy()
    if ref is None: ref=o
    print(f"levels {lv:2d}: {dt*1e3:.3f} ms/step  spa {sp/8:.3f} ms  slow-path variants/step {slow/8:.1f}  max|diff| vs 12 levels {np.nanmax(np.abs(o-ref)):.2e}")
