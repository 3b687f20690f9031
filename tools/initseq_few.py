"""initseq on the first 1 / 3 / 8 columns of a resident 1e8-row AR(1) series, centred and not (MCMCHIP_TRACE=1 prints the rounds).
usage: python tools/initseq_few.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_secondary  # noqa: E402
from mcmc_amd import api  # noqa: E402

api.init(0)
n = 100_000_256
x = bench_secondary.ar1_columns(n, 8, torch, torch.device("cuda:0"))
for p in (1, 3, 8):
    for centred in (True, False):
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            r = api.initseq(x[:p], centred=centred)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("p=%d centred=%s: %.2f ms (first call %.2f), lag pairs %s, %s launches" % (p, centred, min(ts[1:]), ts[0], [len(c["Gamma.pos"]) for c in r], api.last_kernel_time()[2]), flush=True)
