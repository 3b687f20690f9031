"""Kernel time of the stream path for the closed-form descriptors (multivariate normal, normal mixture), 4096 chains.
usage (GPU box): python tools/stream_kinds.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api
api.init()
rng = np.random.default_rng(0)
C, nb = 4096, 2000
for d in (2, 8, 33):
    A = rng.standard_normal((d, d))
    tg = api.Target(abi.T_MVNORM, d, par=np.concatenate([rng.standard_normal(d), (A @ A.T + d * np.eye(d)).ravel()]))
    init = rng.standard_normal((C, d))
    api.metrop(tg, init, 2, 1, 1, scale=0.5 / np.sqrt(d), seed=1)
    api.metrop(tg, init, nb, 1, 1, scale=0.5 / np.sqrt(d), seed=1)
    print("mvnorm d=%d: %.2f ms, %.3g chain-it/s" % (d, api.last_kernel_time()[1], C * nb / api.last_kernel_time()[1] * 1e3), flush=True)
    tg = api.Target(abi.T_NORMMIX, d, par=abi.normmix_par([0.5, 0.3, 0.2], rng.standard_normal((3, d)) * 2, [1.0, 0.5, 2.0]))
    api.metrop(tg, init, 2, 1, 1, scale=1.0 / np.sqrt(d), seed=1)
    api.metrop(tg, init, nb, 1, 1, scale=1.0 / np.sqrt(d), seed=1)
    print("normmix d=%d: %.2f ms, %.3g chain-it/s" % (d, api.last_kernel_time()[1], C * nb / api.last_kernel_time()[1] * 1e3), flush=True)
