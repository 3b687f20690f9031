#!/usr/bin/env python3
"""C5 initseq alone at full size (n x p AR(1) columns generated on the device), several timed repetitions; the series
generation is tests/secondary_bench.py's.  usage: python tools/c5_initseq_only.py [n] [p] [reps]"""
import json
import os
import sys
import time

import numpy as np
import torch  # before libmcmchip.so: both bring a HIP runtime, torch's has to be the one that initialises first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api  # noqa: E402


def gen(n, p):
    dev = torch.device("cuda", 0)
    B = 512
    x = torch.empty((p, n), dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    s_idx = torch.arange(B, dtype=torch.float64, device=dev)
    for c in range(p):
        rho = 0.5 + 0.49 * c / max(p - 1, 1)
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
    x -= x.mean(dim=1, keepdim=True)
    torch.cuda.synchronize()
    return x


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000256
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    api.init()
    L = api.load_library()
    x = gen(n, p)
    cap = 8192
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(p) for _ in range(4))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    secs = []
    for rep in range(reps):
        t0 = time.perf_counter()
        rc = L.mcmchip_initseq(x.data_ptr(), n, p, 1, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data,
                               Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
        secs.append(time.perf_counter() - t0)
        assert rc == 0, L.mcmchip_last_error()
    lags = 2 * n_out
    print(json.dumps({"config": "C5 initseq n=%d p=%d" % (n, p), "seconds": secs, "useful_lag_columns": int(lags.sum()),
                      "useful_TFLOPs_best": float(2.0 * (lags * n).sum() / min(secs) / 1e12),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("MCMCHIP_")},
                      "checksum": float(vc.sum())}))
