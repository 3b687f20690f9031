"""mcmchip_series_upload of a 1.6 GB column-major host matrix in a process without torch (the system's HIP runtime):
GB/s of the pinned-ring upload.  usage: python tools/upload_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import api  # noqa: E402

api.init(0)
x = np.asfortranarray(np.random.default_rng(0).standard_normal((25_000_000, 8)))
for rep in range(4):
    t0 = time.perf_counter()
    ds = api.upload_series(x)
    el = time.perf_counter() - t0
    print("upload %.2f GB in %.1f ms: %.1f GB/s" % (x.nbytes / 1e9, el * 1e3, x.nbytes / el / 1e9), flush=True)
    ds.free()
