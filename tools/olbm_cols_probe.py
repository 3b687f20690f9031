"""olbm over the first p columns of one resident AR(1) series, p from a list: which kernel served the call and how long
it took.  usage (GPU box): python tools/olbm_cols_probe.py n len p1,p2,...   (MCMCHIP_OLBM_LANE / _SMALL / _COMB steer the
routing; they are read once per process)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mcmc_amd import api  # noqa: E402

n, blen = int(sys.argv[1]), int(sys.argv[2])
plist = [int(v) for v in sys.argv[3].split(",")]
pmax = max(plist)
api.init()
L = api.load_library()
dev = torch.device("cuda", 0)
B = 512
n -= n % B
x = torch.empty((pmax, n), dtype=torch.float64, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(5)
s_idx = torch.arange(B, dtype=torch.float64, device=dev)
for c in range(pmax):
    rho = 0.5 + 0.49 * c / max(pmax - 1, 1)
    e = torch.randn((n // B, B), dtype=torch.float64, device=dev, generator=g)
    w = rho ** s_idx
    loc = torch.cumsum(e / w, dim=1) * w
    last = loc[:, -1].clone()
    carry = torch.zeros_like(last)
    carry[1:] = last[:-1]
    carry[2:] += (rho ** B) * last[:-2]
    loc += carry[:, None] * (rho * w)[None, :]
    x[c] = loc.reshape(-1)
    del e, loc
x += 3.0
torch.cuda.synchronize()
print("env LANE=%s SMALL=%s COMB=%s" % tuple(os.environ.get("MCMCHIP_OLBM_" + k) for k in ("LANE", "SMALL", "COMB")), flush=True)
for p in plist:
    mean, var = np.zeros(p), np.empty((p, p))
    best, kn = 1e9, ""
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.mcmchip_olbm(x.data_ptr(), n, p, blen, mean.ctypes.data, var.ctypes.data, 0, 1)
        el = time.perf_counter() - t0
        assert rc == 0, L.mcmchip_last_error()
        kn, kms, _ = api.last_kernel_time()
        best = min(best, kms)
    r0 = 0.5
    print("n=%d len=%d p=%3d  %-20s kernel %8.3f ms  call %8.3f ms  input %6.2f TB/s  var00/theory %.4f" % (
        n, blen, p, kn, best, el * 1e3, 8.0 * n * p / (best * 1e-3) / 1e12,
        var[0, 0] * n / ((1 / (1 - r0 * r0)) * (1 + r0) / (1 - r0))), flush=True)
