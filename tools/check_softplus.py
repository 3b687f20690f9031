"""Accuracy check of mcmc_amd/csrc/softplus.h compiled for the host (no GPU needed).
usage: python tools/check_softplus.py"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    so = os.path.join(tempfile.gettempdir(), "libsp_check.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "check_softplus.cpp")])
    L = ctypes.CDLL(so)
    L.softplus_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    L.softplus_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    return L


def run(a):
    L = load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    out, ref = np.empty_like(a), np.empty_like(a)
    L.softplus_host(a.ctypes.data, out.ctypes.data, a.size)
    L.softplus_ref(a.ctypes.data, ref.ctypes.data, None, a.size)
    return out, ref


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-36, 36, 3_000_000), rng.uniform(0, 1, 500000), rng.uniform(36, 1e6, 100000),
                        [0.0, -0.0, 1e-300, 1e-17, 35.999, 36.0, 36.01, 1e9, 1e12, 1e30],
                        np.arange(0, 2035) / 56.5, (np.arange(0, 2035) + 0.5) / 56.5])
    out, ref = run(a)
    rel = np.abs(out - ref) / np.abs(ref) / 2.0 ** -53
    print("max err %.3f ulp at a=%r; mean %.3f ulp" % (rel.max(), a[rel.argmax()], rel.mean()))
    o2, _ = run(np.array([np.inf, -np.inf, np.nan]))
    print("inf, -inf, nan ->", o2)
    sys.exit(0 if rel.max() < 4.5 and o2[0] == np.inf and o2[1] == np.inf and np.isnan(o2[2]) else 1)
